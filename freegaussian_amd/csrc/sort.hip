// K4: C-ABI entry points of the stable LSD radix sort (kernels in radix_sort.h).
#include "radix_sort.h"

extern "C" size_t fg_sort_workspace_bytes(int64_t n) { return fg_sort::workspace_bytes<uint64_t>(n); }

extern "C" int fg_sort_pairs(int64_t n, int64_t* keys, int32_t* vals, int end_bit, void* workspace,
                             size_t workspace_bytes, fg_stream_t stream) {
  if (n < 0 || end_bit < 0 || end_bit > 64) return FG_ERR_INVALID_ARG;
  if (n <= 1 || end_bit == 0) return FG_OK;
  if (!keys || !vals || !workspace) return FG_ERR_INVALID_ARG;
  return fg_sort::sort_pairs<uint64_t>(n, reinterpret_cast<uint64_t*>(keys), reinterpret_cast<uint32_t*>(vals),
                                       end_bit, workspace, workspace_bytes, fg_hip_stream(stream));
}

extern "C" size_t fg_sort32_workspace_bytes(int64_t n) { return fg_sort::workspace_bytes<uint32_t>(n); }

extern "C" int fg_sort_pairs32(int64_t n, uint32_t* keys, int32_t* vals, int end_bit, void* workspace,
                               size_t workspace_bytes, fg_stream_t stream) {
  if (n < 0 || end_bit < 0 || end_bit > 32) return FG_ERR_INVALID_ARG;
  if (n <= 1 || end_bit == 0) return FG_OK;
  if (!keys || !vals || !workspace) return FG_ERR_INVALID_ARG;
  return fg_sort::sort_pairs<uint32_t>(n, keys, reinterpret_cast<uint32_t*>(vals), end_bit, workspace,
                                       workspace_bytes, fg_hip_stream(stream));
}
