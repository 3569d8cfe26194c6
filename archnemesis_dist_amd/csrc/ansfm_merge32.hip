// ansfm_merge32.hip -- translation unit of k_ck_overlap32 (ansfm_merge32.hip.h): the instantiations and their launcher.
#include "ansfm_merge32.hip.h"
#include "ansfm_merge32_launch.h"

namespace ansfm {

unsigned overlap32_lds_bytes(int G, bool delg_f32) { return m32_lds_bytes(G, delg_f32); }

hipError_t launch_overlap32(const OverlapParams &p, bool from_k, int list_len, unsigned grid, hipStream_t stream)
{
    const size_t lds = m32_lds_bytes(p.G, p.delg_f32 != 0);
#define LAUNCH_M32(D, FK, W32) \
    hipLaunchKernelGGL((k_ck_overlap32<D, FK, W32>), dim3(grid), dim3(kWave), lds, stream, p)
#define LAUNCH_M32_D(D)                                                             \
    do {                                                                            \
        if (from_k) { if (p.delg_f32) LAUNCH_M32(D, true, true); else LAUNCH_M32(D, true, false); }     \
        else { if (p.delg_f32) LAUNCH_M32(D, false, true); else LAUNCH_M32(D, false, false); }          \
    } while (0)
    switch (list_len) {
        case 8: LAUNCH_M32_D(8); break;
        case 10: LAUNCH_M32_D(10); break;
        case 16: LAUNCH_M32_D(16); break;
        case 20: LAUNCH_M32_D(20); break;
        default: LAUNCH_M32_D(32); break;
    }
#undef LAUNCH_M32_D
#undef LAUNCH_M32
    return hipGetLastError();
}

}  // namespace ansfm
