"""Drop-in for the reference's EmbeddingService (src/voitta/services/embedding.py): same class,
method names, e5 prefix rules, batch argument and return types (plain Python lists).

``SentenceTransformer(model_name, device)`` (embedding.py:40) is replaced by NativeSentenceEncoder:
the checkpoint directory is read the way sentence-transformers reads it (config.json, modules.json,
1_Pooling/config.json, sentence_bert_config.json, tokenizer.json / vocab.txt, model.safetensors),
PyTorch only loads and holds the weights, the HF ``tokenizers`` library (the one
sentence-transformers itself uses) produces WordPiece ids, and the forward pass runs in the HIP
engine (vr_encode). There is no CPU path: EMBEDDING_DEVICE=cpu raises."""
from __future__ import annotations

import json
import logging
import os

import numpy as np

from . import deferred as _deferred
from . import encoder as _enc
from .config import get_settings
from .store_registry import get_engine
from .wordpiece import WordPieceTokenizer

logger = logging.getLogger(__name__)


class NativeSentenceEncoder:
    """The object behind ``EmbeddingService.model``: ``encode(texts)`` like SentenceTransformer's."""

    def __init__(self, engine, desc: _enc.BertDesc, state: dict, tokenizer, max_seq_length: int):
        self.engine = engine
        self.desc = desc
        self.tokenizer = tokenizer
        self.max_seq_length = min(max_seq_length, desc.max_pos)
        if isinstance(tokenizer, WordPieceTokenizer):
            tokenizer.max_length = self.max_seq_length
        else:
            self.tokenizer.no_padding()
            self.tokenizer.enable_truncation(max_length=self.max_seq_length)
        _enc.load_encoder(engine, desc, state)

    # ---- loading ---------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, path: str, engine=None) -> "NativeSentenceEncoder":
        if not os.path.isdir(path):
            raise FileNotFoundError(
                f"EMBEDDING_MODEL='{path}' is not a local checkpoint directory (hub names cannot be "
                "downloaded here: there is no network). Point it at a sentence-transformers / HF BERT directory.")
        cfg = json.load(open(os.path.join(path, "config.json")))
        if cfg.get("model_type", "bert") != "bert":
            raise ValueError(f"unsupported model_type {cfg.get('model_type')}: BERT-family encoders only")
        if cfg.get("hidden_act", "gelu") != "gelu" or cfg.get("position_embedding_type", "absolute") != "absolute":
            raise ValueError("only exact-erf GELU and absolute position embeddings are implemented")
        pooling, normalize, max_seq = "mean", False, int(cfg["max_position_embeddings"])
        mod_path = os.path.join(path, "modules.json")
        if os.path.exists(mod_path):
            for m in json.load(open(mod_path)):
                kind = m.get("type", "")
                if kind.endswith("Pooling"):
                    pc = json.load(open(os.path.join(path, m["path"], "config.json")))
                    if pc.get("pooling_mode_cls_token"):
                        pooling = "cls"
                    elif not pc.get("pooling_mode_mean_tokens", True):
                        raise ValueError("only CLS and mean pooling are implemented")
                elif kind.endswith("Normalize"):
                    normalize = True
        sb = os.path.join(path, "sentence_bert_config.json")
        if os.path.exists(sb):
            max_seq = int(json.load(open(sb)).get("max_seq_length", max_seq))
        desc = _enc.BertDesc(layers=cfg["num_hidden_layers"], hidden=cfg["hidden_size"],
                             heads=cfg["num_attention_heads"], intermediate=cfg["intermediate_size"],
                             vocab=cfg["vocab_size"], max_pos=cfg["max_position_embeddings"],
                             type_vocab=cfg.get("type_vocab_size", 2), pooling=pooling, normalize=normalize,
                             eps=cfg.get("layer_norm_eps", 1e-12),
                             # f16 (default): f16 MFMA operands, f32 accumulate, |1 - cos| < 1e-6 vs f64
                             # (north_star allows 1e-4); f16x3: (hi, lo) f16 operands, three passes,
                             # |1 - cos| ~5e-8, 1.8x slower; f32: the f32-input MFMA, 5x slower
                             precision=os.environ.get("VOITTA_ENCODER_PRECISION", "f16"))
        return cls(engine or get_engine(), desc, cls._load_weights(path), cls._load_tokenizer(path, cfg), max_seq)

    @staticmethod
    def _load_weights(path: str) -> dict:
        st = os.path.join(path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file

            return load_file(st)
        pt = os.path.join(path, "pytorch_model.bin")
        if os.path.exists(pt):
            import torch

            return torch.load(pt, map_location="cpu", weights_only=True)
        raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {path}")

    @staticmethod
    def _load_tokenizer(path: str, cfg: dict):
        """The native WordPiece (csrc/wordpiece.cpp) for plain BERT tokenizers; anything else that a
        tokenizer.json may describe goes through the HF `tokenizers` library."""
        tj = os.path.join(path, "tokenizer.json")
        if os.path.exists(os.path.join(path, "vocab.txt")) or os.path.exists(tj):
            try:
                if os.path.exists(tj):
                    spec = json.load(open(tj, encoding="utf-8"))
                    plain = (spec["model"]["type"] == "WordPiece"
                             and (spec.get("normalizer") or {}).get("type") == "BertNormalizer"
                             and (spec.get("pre_tokenizer") or {}).get("type") == "BertPreTokenizer"
                             and spec["model"].get("continuing_subword_prefix", "##") == "##"
                             and spec["model"].get("max_input_chars_per_word", 100) == 100)
                    if not plain:
                        raise ValueError("not a plain BERT WordPiece pipeline")
                return WordPieceTokenizer.from_pretrained(path)
            except (ValueError, KeyError) as e:
                logger.info("native WordPiece not applicable (%s); using the tokenizers library", e)
        from tokenizers import Tokenizer

        if os.path.exists(tj):
            return Tokenizer.from_file(tj)
        vocab = os.path.join(path, "vocab.txt")
        if not os.path.exists(vocab):
            raise FileNotFoundError(f"no tokenizer.json / vocab.txt under {path}")
        lower = True
        tc = os.path.join(path, "tokenizer_config.json")
        if os.path.exists(tc):
            lower = bool(json.load(open(tc)).get("do_lower_case", True))
        return build_wordpiece_tokenizer([l.rstrip("\n") for l in open(vocab, encoding="utf-8")], lower)

    # ---- SentenceTransformer.encode -------------------------------------------------------------
    def tokenize(self, texts: list[str]):
        if isinstance(self.tokenizer, WordPieceTokenizer):
            ids, off = self.tokenizer.encode_batch(list(texts))
            return ids, off.astype(np.int32)
        encs = self.tokenizer.encode_batch(list(texts))
        lens = [len(e.ids) for e in encs]
        off = np.zeros(len(texts) + 1, np.int32)
        off[1:] = np.cumsum(lens)
        ids = np.fromiter((t for e in encs for t in e.ids), dtype=np.int32, count=int(off[-1]))
        return ids, off

    def encode(self, sentences, batch_size: int = 32, convert_to_numpy: bool = True,
               show_progress_bar: bool = False, **_ignored):
        """str -> (D,) array, list[str] -> (n, D) array. batch_size shaped only the padding of the
        reference's batches; the packed GPU layout has none, so it is accepted and ignored."""
        single = isinstance(sentences, str)
        texts = [sentences] if single else list(sentences)
        if not texts:
            return np.zeros((0, self.desc.hidden), np.float32)
        ids, off = self.tokenize(texts)
        out = _enc.encode(self.engine, ids, off)
        return out[0] if single else out


def build_wordpiece_tokenizer(vocab: list[str], lowercase: bool = True):
    """BertTokenizerFast equivalent from a vocab.txt: BertNormalizer + BertPreTokenizer + WordPiece
    ('##', 100-char word limit) + '[CLS] $A [SEP]' post-processing."""
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors

    v = {t: i for i, t in enumerate(vocab)}
    tok = Tokenizer(models.WordPiece(v, unk_token="[UNK]", max_input_chars_per_word=100))
    tok.normalizer = normalizers.BertNormalizer(clean_text=True, handle_chinese_chars=True, strip_accents=None,
                                                lowercase=lowercase)
    tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tok.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]",
                                                       special_tokens=[("[CLS]", v["[CLS]"]), ("[SEP]", v["[SEP]"])])
    return tok


class EmbeddingService:
    """Service for generating text embeddings on the GPU (reference: embedding.py:14-86)."""

    def __init__(self, model_name: str | None = None):
        settings = get_settings()
        self.model_name = model_name or settings.embedding_model
        self.dimension = settings.embedding_dimension  # taken from settings, not the model (embedding.py:20)
        self._model: NativeSentenceEncoder | None = None

    @property
    def model(self) -> NativeSentenceEncoder:
        """Lazy load the model (embedding.py:23-42)."""
        if self._model is None:
            settings = get_settings()
            logger.info(f"Loading embedding model: {self.model_name}")
            device_setting = settings.embedding_device.lower()
            if device_setting == "cpu":
                raise RuntimeError("EMBEDDING_DEVICE=cpu: the native embedding service has no CPU path "
                                   "(use the reference's sentence-transformers service instead)")
            self._model = NativeSentenceEncoder.from_pretrained(self.model_name)
            logger.info("Model loaded successfully on the MI355X engine")
        return self._model

    def embed_text(self, text: str) -> list[float]:
        if "e5" in self.model_name.lower():  # embedding.py:50-51
            text = f"passage: {text}"
        return self.model.encode(text, convert_to_numpy=True).tolist()

    def embed_texts(self, texts: list[str], batch_size: int = 32) -> list[list[float]]:
        if not texts:
            return []
        if "e5" in self.model_name.lower():  # embedding.py:65-66
            texts = [f"passage: {text}" for text in texts]
        if _deferred.enabled():
            # tokenised now, encoded when somebody looks at a number — or, when the list goes to store_chunks
            # untouched, together with thousands of other chunks inside the engine (voitta_rag_amd/deferred.py)
            return _deferred.DeferredEmbeddings(self.model, *self.model.tokenize(list(texts)))
        embeddings = self.model.encode(texts, batch_size=batch_size, convert_to_numpy=True,
                                       show_progress_bar=len(texts) > 100)
        return embeddings.tolist()

    def embed_query(self, query: str) -> list[float]:
        if "e5" in self.model_name.lower():  # embedding.py:82-83
            query = f"query: {query}"
        model = self.model
        if _deferred.enabled() and isinstance(model.tokenizer, WordPieceTokenizer):
            return _deferred.QueryRef(model, query)  # encoded when looked at — or inside the search call it goes to
        return _deferred.QueryEmbedding(model.encode(query, convert_to_numpy=True))


_embedding_service: EmbeddingService | None = None


def get_embedding_service() -> EmbeddingService:
    global _embedding_service
    if _embedding_service is None:
        _embedding_service = EmbeddingService()
    return _embedding_service
