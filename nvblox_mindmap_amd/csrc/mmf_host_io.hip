// mmf_host_io.hip -- host-only: the two reads a training sample costs the loader, written so that the ONLY copy is from the
// page cache into the caller's buffer (a row of a pinned batch buffer: data_loading/pinned_loader.py).
//
// The reference's loader (mindmap/data_loading/dataset.py:410-415,457-468) inflates two PNGs and decompresses + unpickles the
// whole feature mesh of a frame (10-14 k vertices x 768 f16 channels, ~18 MB) to keep 2 048 rows
// (data_loading/sample_transformer.py:150-186), then collates per-sample tensors into a batch, then pins it: three copies of
// every byte and 5 ms of CPU per sample.  Here the raw copies of io/vertex_cache.py are read in place:
//   * mmf_host_read_file_at      pread() of an image's pixel block straight into its row of the batch buffer;
//   * mmf_host_sample_vertex_file  mmap() of the raw vertex-feature file, the selected rows copied in ascending file order
//                                  (the page cache's readahead / fault-around see a forward walk), munmap().
// Both release nothing and hold no state; ctypes drops the interpreter lock around them, so a small thread pool scales.
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/mmfusion.h"
#include "mmf_api_internal.h"

using mmf_host::fail;

extern "C" int mmf_host_read_file_at(const char* path, int64_t offset, void* dst_host, int64_t nbytes) {
  if (!path || !dst_host || offset < 0 || nbytes < 0) return fail(MMF_ERR_INVALID_ARG, "mmf_host_read_file_at: bad argument");
  int fd = open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return fail(MMF_ERR_INVALID_ARG, std::string("mmf_host_read_file_at: cannot open ") + path + ": " + strerror(errno));
  int64_t done = 0;
  while (done < nbytes) {
    ssize_t r = pread(fd, static_cast<char*>(dst_host) + done, static_cast<size_t>(nbytes - done), offset + done);
    if (r < 0 && errno == EINTR) continue;
    if (r <= 0) break;
    done += r;
  }
  close(fd);
  if (done != nbytes) return fail(MMF_ERR_INVALID_ARG, std::string("mmf_host_read_file_at: short read of ") + path);
  return MMF_OK;
}

extern "C" int mmf_host_sample_vertex_file(const char* path, int64_t off_vertices, int64_t off_features, int64_t num_vertices,
                                           int64_t channels, const int64_t* rows_host, int64_t n_rows, void* vertices_f16_out_host,
                                           void* features_f16_out_host) {
  if (!path || off_vertices < 0 || off_features < off_vertices || num_vertices < 0 || channels <= 0 || n_rows < 0 ||
      (n_rows > 0 && (!rows_host || !vertices_f16_out_host || !features_f16_out_host)))
    return fail(MMF_ERR_INVALID_ARG, "mmf_host_sample_vertex_file: bad argument");
  if (n_rows == 0) return MMF_OK;
  for (int64_t i = 0; i < n_rows; i++)
    if (rows_host[i] < 0 || rows_host[i] >= num_vertices) return fail(MMF_ERR_INVALID_ARG, "mmf_host_sample_vertex_file: row index out of range");
  int fd = open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return fail(MMF_ERR_INVALID_ARG, std::string("mmf_host_sample_vertex_file: cannot open ") + path + ": " + strerror(errno));
  struct stat st;
  const int64_t row_bytes = channels * 2;
  if (fstat(fd, &st) != 0 || st.st_size < off_features + num_vertices * row_bytes || off_features < off_vertices + num_vertices * 6) {
    close(fd);
    return fail(MMF_ERR_INVALID_ARG, std::string("mmf_host_sample_vertex_file: ") + path + " is smaller than its header says");
  }
  void* map = mmap(nullptr, static_cast<size_t>(st.st_size), PROT_READ, MAP_SHARED, fd, 0);
  close(fd);
  if (map == MAP_FAILED) return fail(MMF_ERR_INVALID_ARG, std::string("mmf_host_sample_vertex_file: mmap of ") + path + " failed");
  const char* base = static_cast<const char*>(map);
  // ascending file order; the output position of every row is kept
  std::vector<std::pair<int64_t, int64_t>> order(static_cast<size_t>(n_rows));
  for (int64_t i = 0; i < n_rows; i++) order[static_cast<size_t>(i)] = {rows_host[i], i};
  std::sort(order.begin(), order.end());
  const char* v = base + off_vertices;
  const char* f = base + off_features;
  char* vo = static_cast<char*>(vertices_f16_out_host);
  char* fo = static_cast<char*>(features_f16_out_host);
  for (const auto& [row, pos] : order) {
    memcpy(vo + pos * 6, v + row * 6, 6);
    memcpy(fo + pos * row_bytes, f + row * row_bytes, static_cast<size_t>(row_bytes));
  }
  munmap(map, static_cast<size_t>(st.st_size));
  return MMF_OK;
}
