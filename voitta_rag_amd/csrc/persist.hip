// On-disk image of an engine's index. The reference's state survives a
// restart in Qdrant's volume (reference: docker-compose.yml:8-9, services/vector_store.py:75-115
// re-attaches to the collection); an HBM-resident index needs its own file. One file holds
// everything the device owns: the tiled f32 corpus, payload columns, tombstones, the SELL sparse
// index and the document-frequency table. The f16 shadow is derived data and is rebuilt on load.
// Payload text, point ids and the folder dictionaries are the host's (vector_store.py writes them
// next to this file).
//
// Layout, little endian: FileHeader, then the sections in the order of `Section`, each padded to
// 64 bytes. Sections are raw device images, so a file is tied to the layout version in the header.

#include "engine_internal.h"

#include <fcntl.h>
#include <unistd.h>

#include <cstdio>
#include <string>

namespace vr {

namespace {

constexpr uint32_t kFileVersion = 3;  // 3: tile_pos() layout of the f32 corpus blocks (engine_internal.h)
constexpr size_t kChunk = size_t(32) << 20;  // staging buffer (pinned)

struct FileHeader {
  char magic[8];  // "VRINDEX\0"
  uint32_t version;
  uint32_t dim;
  int64_t n_rows;
  int64_t n_live;
  int32_t max_folder_id;
  int32_t max_index_folder_id;
  int64_t n_slices;
  int64_t sp_used;
  int64_t n_sparse_points;
  int64_t df_cap;
  int64_t df_distinct;
  uint64_t payload_bytes;  // everything after the header
  uint64_t checksum;       // of the payload, see mix()
};
static_assert(sizeof(FileHeader) == 96, "file header layout");

inline uint64_t mix(uint64_t h, const void* data, size_t n) {  // word-wise multiply-xorshift
  const unsigned char* p = static_cast<const unsigned char*>(data);
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    memcpy(&w, p + i, 8);
    h = (h ^ w) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
  }
  uint64_t tail = 0;
  if (i < n) memcpy(&tail, p + i, n - i);
  h = (h ^ tail ^ static_cast<uint64_t>(n)) * 0xD6E8FEB86659FD93ull;
  return h ^ (h >> 32);
}

struct Io {
  FILE* f = nullptr;
  void* stage = nullptr;  // pinned
  uint64_t sum = 0x243F6A8885A308D3ull;
  uint64_t bytes = 0;
  ~Io() {
    if (f) fclose(f);
    if (stage) (void)hipHostFree(stage);
  }
};

int pad64(Io& io, bool writing) {
  static const char zeros[64] = {0};
  const size_t r = static_cast<size_t>(io.bytes % 64);
  if (r == 0) return 0;
  char buf[64];
  if (writing) {
    VR_CHECK(fwrite(zeros, 1, 64 - r, io.f) == 64 - r, "short write");
    io.sum = mix(io.sum, zeros, 64 - r);
  } else {
    VR_CHECK(fread(buf, 1, 64 - r, io.f) == 64 - r, "index file truncated");
    io.sum = mix(io.sum, buf, 64 - r);
  }
  io.bytes += 64 - r;
  return 0;
}

int write_host(Io& io, const void* p, size_t n) {
  if (n) VR_CHECK(fwrite(p, 1, n, io.f) == n, "short write (disk full?)");
  io.sum = mix(io.sum, p, n);
  io.bytes += n;
  return pad64(io, true);
}

int write_dev(vr_engine* e, Io& io, const void* dev, size_t n) {
  for (size_t at = 0; at < n; at += kChunk) {
    const size_t m = std::min(kChunk, n - at);
    VR_HIP(hipMemcpyAsync(io.stage, static_cast<const char*>(dev) + at, m, hipMemcpyDeviceToHost, e->stream));
    VR_HIP(hipStreamSynchronize(e->stream));
    VR_CHECK(fwrite(io.stage, 1, m, io.f) == m, "short write (disk full?)");
    io.sum = mix(io.sum, io.stage, m);
  }
  io.bytes += n;
  return pad64(io, true);
}

int read_host(Io& io, void* p, size_t n) {
  if (n) VR_CHECK(fread(p, 1, n, io.f) == n, "index file truncated");
  io.sum = mix(io.sum, p, n);
  io.bytes += n;
  return pad64(io, false);
}

int read_dev(vr_engine* e, Io& io, void* dev, size_t n) {
  for (size_t at = 0; at < n; at += kChunk) {
    const size_t m = std::min(kChunk, n - at);
    VR_CHECK(fread(io.stage, 1, m, io.f) == m, "index file truncated");
    io.sum = mix(io.sum, io.stage, m);
    VR_HIP(hipMemcpyAsync(static_cast<char*>(dev) + at, io.stage, m, hipMemcpyHostToDevice, e->stream));
    VR_HIP(hipStreamSynchronize(e->stream));  // the staging buffer is reused
  }
  io.bytes += n;
  return pad64(io, false);
}

}  // namespace

int engine_save(vr_engine* e, const char* path) {
  VR_HIP(hipStreamSynchronize(e->stream));
  Io io;
  const std::string tmp = std::string(path) + ".tmp";
  io.f = fopen(tmp.c_str(), "wb");
  VR_CHECK(io.f != nullptr, "cannot open %s for writing", tmp.c_str());
  VR_HIP(hipHostMalloc(&io.stage, kChunk, hipHostMallocDefault));
  FileHeader h;
  memset(&h, 0, sizeof(h));
  memcpy(h.magic, "VRINDEX", 8);
  h.version = kFileVersion;
  h.dim = static_cast<uint32_t>(e->dim);
  h.n_rows = e->n_rows;
  h.n_live = e->n_live;
  h.max_folder_id = e->max_folder_id;
  h.max_index_folder_id = e->max_index_folder_id;
  h.n_slices = static_cast<int64_t>(e->slices_host.size());
  h.sp_used = e->sp_used;
  h.n_sparse_points = e->n_sparse_points;
  h.df_cap = e->df_cap;
  if (e->df_distinct) {
    int32_t d = 0;
    VR_HIP(hipMemcpy(&d, e->df_distinct, sizeof(d), hipMemcpyDeviceToHost));
    h.df_distinct = d;
  }
  VR_CHECK(fwrite(&h, 1, sizeof(h), io.f) == sizeof(h), "short write");  // rewritten at the end
  const size_t n = static_cast<size_t>(e->n_rows);
  const size_t tiles = (n + kTileRows - 1) / kTileRows;
  VR_TRY(write_dev(e, io, e->corpus.p, tiles * kTileRows * e->dim * sizeof(float)));
  VR_TRY(write_dev(e, io, e->live.p, n));
  VR_TRY(write_dev(e, io, e->folder.p, n * 4));
  VR_TRY(write_dev(e, io, e->index_folder.p, n * 4));
  VR_TRY(write_dev(e, io, e->created.p, n * 8));
  VR_TRY(write_dev(e, io, e->modified.p, n * 8));
  VR_TRY(write_dev(e, io, e->row_slice.p, n * 4));
  VR_TRY(write_host(io, e->slices_host.data(), e->slices_host.size() * sizeof(SliceDesc)));
  VR_TRY(write_dev(e, io, e->sp_idx.p, static_cast<size_t>(e->sp_used) * 4));
  VR_TRY(write_dev(e, io, e->sp_val.p, static_cast<size_t>(e->sp_used) * 4));
  VR_TRY(write_dev(e, io, e->df_keys.p, static_cast<size_t>(e->df_cap) * 4));
  VR_TRY(write_dev(e, io, e->df_cnt.p, static_cast<size_t>(e->df_cap) * 4));
  h.payload_bytes = io.bytes;
  h.checksum = io.sum;
  VR_CHECK(fseek(io.f, 0, SEEK_SET) == 0 && fwrite(&h, 1, sizeof(h), io.f) == sizeof(h), "cannot finish %s", tmp.c_str());
  // durable before it becomes visible: data to disk, then the rename, then the directory entry
  VR_CHECK(fflush(io.f) == 0 && fsync(fileno(io.f)) == 0, "flush of %s failed", tmp.c_str());
  VR_CHECK(fclose(io.f) == 0, "closing %s failed", tmp.c_str());
  io.f = nullptr;
  VR_CHECK(rename(tmp.c_str(), path) == 0, "cannot rename %s to %s", tmp.c_str(), path);  // atomic replace
  {
    std::string dir(path);
    const size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? "." : (slash == 0 ? "/" : dir.substr(0, slash));
    const int dfd = open(dir.c_str(), O_RDONLY | O_DIRECTORY);
    if (dfd >= 0) {
      (void)fsync(dfd);
      close(dfd);
    }
  }
  return 0;
}

int engine_load(vr_engine* e, const char* path) {
  VR_CHECK(e->n_rows == 0 && e->slices_host.empty(), "vr_load needs an empty engine (%lld rows present)",
           static_cast<long long>(e->n_rows));
  Io io;
  io.f = fopen(path, "rb");
  VR_CHECK(io.f != nullptr, "cannot open %s", path);
  FileHeader h;
  VR_CHECK(fread(&h, 1, sizeof(h), io.f) == sizeof(h), "%s: no header", path);
  VR_CHECK(memcmp(h.magic, "VRINDEX", 8) == 0, "%s is not an index file", path);
  VR_CHECK(h.version == kFileVersion, "%s: layout version %u, this library reads %u", path, h.version, kFileVersion);
  VR_CHECK(static_cast<int>(h.dim) == e->dim, "%s holds %u-dimensional vectors, the engine is %d-dimensional", path,
           h.dim, e->dim);
  VR_CHECK(h.n_rows >= 0 && h.n_live >= 0 && h.n_live <= h.n_rows && h.n_slices >= 0 && h.sp_used >= 0 &&
               h.df_cap >= 0 && (h.df_cap & (h.df_cap - 1)) == 0 && h.n_rows < (int64_t(1) << 31),
           "%s: implausible header", path);
  VR_HIP(hipHostMalloc(&io.stage, kChunk, hipHostMallocDefault));
  const size_t n = static_cast<size_t>(h.n_rows);
  const size_t tiles = (n + kTileRows - 1) / kTileRows;
  VR_TRY(ensure_rows(e, std::max<int64_t>(h.n_rows, 1)));
  VR_TRY(read_dev(e, io, e->corpus.p, tiles * kTileRows * e->dim * sizeof(float)));
  VR_TRY(read_dev(e, io, e->live.p, n));
  VR_TRY(read_dev(e, io, e->folder.p, n * 4));
  VR_TRY(read_dev(e, io, e->index_folder.p, n * 4));
  VR_TRY(read_dev(e, io, e->created.p, n * 8));
  VR_TRY(read_dev(e, io, e->modified.p, n * 8));
  VR_TRY(read_dev(e, io, e->row_slice.p, n * 4));
  std::vector<SliceDesc> slices(static_cast<size_t>(h.n_slices));
  VR_TRY(read_host(io, slices.data(), slices.size() * sizeof(SliceDesc)));
  VR_TRY(e->sp_idx.grow(std::max<int64_t>(h.sp_used, 1), 0, e->stream));
  VR_TRY(e->sp_val.grow(std::max<int64_t>(h.sp_used, 1), 0, e->stream));
  VR_TRY(read_dev(e, io, e->sp_idx.p, static_cast<size_t>(h.sp_used) * 4));
  VR_TRY(read_dev(e, io, e->sp_val.p, static_cast<size_t>(h.sp_used) * 4));
  if (h.df_cap > 0) {
    VR_TRY(e->df_keys.grow(h.df_cap, 0, e->stream));
    VR_TRY(e->df_cnt.grow(h.df_cap, 0, e->stream));
    VR_CHECK(e->df_keys.cap == h.df_cap && e->df_cnt.cap == h.df_cap, "document-frequency table: unexpected capacity");
  }
  VR_TRY(read_dev(e, io, e->df_keys.p, static_cast<size_t>(h.df_cap) * 4));
  VR_TRY(read_dev(e, io, e->df_cnt.p, static_cast<size_t>(h.df_cap) * 4));
  VR_CHECK(io.bytes == h.payload_bytes && io.sum == h.checksum, "%s is corrupt (checksum mismatch)", path);
  // slices must lie inside what was read
  for (const SliceDesc& d : slices)
    VR_CHECK(d.off >= 0 && d.width >= 0 && d.off + static_cast<int64_t>(d.width) * 64 <= h.sp_used &&
                 d.row_base >= 0 && d.nrows >= 0 && d.nrows <= 64 && d.row_base + d.nrows <= h.n_rows,
             "%s: sparse slice out of range", path);
  e->slices_host.swap(slices);
  if (!e->slices_host.empty()) {
    VR_TRY(e->slices.grow(static_cast<int64_t>(e->slices_host.size()), 0, e->stream));
    VR_HIP(hipMemcpyAsync(e->slices.p, e->slices_host.data(), e->slices_host.size() * sizeof(SliceDesc),
                          hipMemcpyHostToDevice, e->stream));
  }
  e->n_slices_dev = static_cast<int64_t>(e->slices_host.size());
  e->sp_used = h.sp_used;
  e->n_sparse_points = h.n_sparse_points;
  e->df_cap = h.df_cap;
  e->df_bound = h.df_distinct;
  if (h.df_cap > 0) {
    if (!e->df_distinct) VR_HIP(hipMalloc(reinterpret_cast<void**>(&e->df_distinct), sizeof(int32_t)));
    const int32_t d = static_cast<int32_t>(h.df_distinct);
    VR_HIP(hipMemcpyAsync(e->df_distinct, &d, sizeof(d), hipMemcpyHostToDevice, e->stream));
  }
  e->n_rows = h.n_rows;
  e->n_live = h.n_live;
  e->max_folder_id = h.max_folder_id;
  e->max_index_folder_id = h.max_index_folder_id;
  e->centre_rows = 0;
  e->centre_checked_rows = 0;
  VR_TRY(prefilter_recentre(e));  // derived data (shadow and its centre): rebuilt, not stored
  VR_TRY(inv_rebuild(e));         // likewise the inverted twin of the sparse slices
  VR_HIP(hipStreamSynchronize(e->stream));
  return 0;
}

}  // namespace vr
