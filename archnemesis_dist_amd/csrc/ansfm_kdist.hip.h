// ansfm_kdist.hip.h -- parameters shared by the k-distribution generator (ansfm_kdist.hip) and the C-ABI (ansfm_api.hip)
#pragma once
#include <stdint.h>

namespace ansfm {

struct KdistParams {
    const double *wavecalc, *kabs;     // [ncalc]
    const int32_t *i0;                 // [nbin] first point of each bin
    const int64_t *off;                // [nbin+1] segment offsets
    const double *wcen;                // [nbin] bin centres (ILS only)
    const int32_t *nfil;               // [nbin] or nullptr: no instrument function (weights 1)
    const double *dfil, *afil;         // [nfilmax][nbin] offsets from the bin centre (ascending) and amplitudes
    const double *g_ord;               // [NG]
    double *keys, *vals;               // segmented (k, w dv) pairs before / after the sort
    double *kout;                      // [nbin][NG]
    double dv;
    int nbin, NG;
};

}  // namespace ansfm

// ansfm_kdist.hip: device pointers in p, result in p.kout; returns a hipError_t as int (0 = success)
extern "C" int ansfm_kdist_run(void *stream, ansfm::KdistParams p, int64_t total);
