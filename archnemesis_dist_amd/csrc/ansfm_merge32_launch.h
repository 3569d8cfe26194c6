// ansfm_merge32_launch.h -- host interface of the 32-bit-key merge kernel's translation unit (ansfm_merge32.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace ansfm {
struct OverlapParams;
// dynamic LDS of one 64-lane block at G g-ordinates
unsigned overlap32_lds_bytes(int G, bool delg_f32);
// p.scratch: [grid][G][64] doubles; list_len = the instantiated list length >= G (8, 10, 16, 20 or 32)
hipError_t launch_overlap32(const OverlapParams &p, bool from_k, int list_len, unsigned grid, hipStream_t stream);
}  // namespace ansfm
