// pt_internal.h -- small helpers shared by the translation units of libmi3pt.so
#pragma once
#include <string>

// Records `msg` as the calling thread's last error and returns `code`.
int pt_set_error(int code, const std::string &msg);

// pt_lbvh.hip: device-side linear BVH over n 112-byte triangle records at d_tris; writes (2n - 1)
// 48-byte node records to the HOST buffer nodes_out.  Returns 0, or -1 with `err` set.
#include <hip/hip_runtime.h>
namespace pt {
int lbvh_build(const void *d_tris, size_t n, void *nodes_out, float *build_ms, hipStream_t stream, std::string &err);
}
