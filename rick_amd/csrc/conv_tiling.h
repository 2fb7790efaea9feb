// Tiling / geometry helpers shared by the implicit-GEMM forward kernels (conv.hip) and the weight-gradient kernels
// (wgrad.hip).
#pragma once
#include "conv_common.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------------------
struct ConvTiling {
    int tw_log2, th_log2;       // position tile = (1<<tw) x (1<<th) x nb images = 128 (igemm) / 64 (wgrad)
    int nb;                     // images per tile
    int ntx, nty, ntn;          // tiles along x, y, image groups
    int dymin, dxmin;
    int PH, PW, NPP;            // patch extents (input pixels), NPP = nb*PH*PW
    int nchunks, ncot;
    int nbe;                    // images per tile actually used (min(nb, N)); patch holds nbe images
    int nsplit, cps;            // igemm split-K over channel chunks: splits, chunks per split
    int pkb;                    // wgrad: bit of the patch pixel index that keys the slot swizzle
    int debug;                  // ablation switches for tools/bench_conv.py (RICK_CONV_DEBUG); 0 in production
    unsigned *tickets;          // igemm split-K: arrival counters, one per output tile (NULL: partial sums only, second-stage launch)
};

// Ablation switches (tools/bench_conv.py) exist only in builds made with -DRICK_ABLATION; the production library never
// reads the environment, so a stray variable cannot change results.
static int ablation_env(const char *name, int dflt) {
#ifdef RICK_ABLATION
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

// A 64-byte page of zeros in device memory: out-of-range staging items load from it, so their registers need no
// zero-select afterwards (4 VALU per item; the conv kernels are bound by the issue of their staging instructions).
static __device__ __attribute__((aligned(64))) float g_zero_page[16];   // (one copy per translation unit)

static int ilog2_ceil(int v) {
    int l = 0;
    while ((1 << l) < v) l++;
    return l;
}

static int make_tiling(const rick_conv_geom *g, int tile_positions, ConvTiling *t) {
    if (g->ntaps < 1 || g->ntaps > RICK_MAX_TAPS) return RICK_EINVAL;
    const int tl = ilog2_ceil(tile_positions);
    int tw = ilog2_ceil(g->GW);
    if (tw > 4) tw = 4;
    if (tw < 2) tw = 2;
    if (tw > tl) tw = tl;
    int th = ilog2_ceil(g->GH);
    if (th > tl - tw) th = tl - tw;
    t->tw_log2 = tw;
    t->th_log2 = th;
    t->nb = tile_positions >> (tw + th);
    t->nbe = t->nb < g->N ? t->nb : g->N;
    t->ntx = cdiv(g->GW, 1 << tw);
    t->nty = cdiv(g->GH, 1 << th);
    t->ntn = cdiv(g->N, t->nbe);
    int dymin = g->dy[0], dymax = g->dy[0], dxmin = g->dx[0], dxmax = g->dx[0];
    for (int i = 1; i < g->ntaps; i++) {
        dymin = g->dy[i] < dymin ? g->dy[i] : dymin;
        dymax = g->dy[i] > dymax ? g->dy[i] : dymax;
        dxmin = g->dx[i] < dxmin ? g->dx[i] : dxmin;
        dxmax = g->dx[i] > dxmax ? g->dx[i] : dxmax;
    }
    t->dymin = dymin;
    t->dxmin = dxmin;
    t->PH = ((1 << th) - 1) * g->is + (dymax - dymin) + 1;
    t->PW = ((1 << tw) - 1) * g->is + (dxmax - dxmin) + 1;
    t->NPP = t->nbe * t->PH * t->PW;
    t->nchunks = cdiv(g->Ci, CV_CK);
    t->ncot = cdiv(g->Co, CV_BM);
    t->nsplit = 1;
    t->cps = t->nchunks;
    t->pkb = (g->is <= 2 && tw == 4) ? 3 : (g->is == 2 && tw == 3) ? 5 : 2;   // (tools/lds_sim.py)
    t->debug = ablation_env("RICK_CONV_DEBUG", 0);
    t->tickets = nullptr;
    return 0;
}


// Patch pixel -> (image-in-tile, row, col) table, built once per block so the staging loops need no
// integer divisions: entry = nbi << 20 | py << 10 | px.
__device__ __forceinline__ void build_patch_table(unsigned *ptab, const ConvTiling &t, int nthr = 256) {
    const int phw = t.PH * t.PW;
    for (int pix = threadIdx.x; pix < t.NPP; pix += nthr) {
        const int nbi = pix / phw;
        const int rem = pix - nbi * phw;
        const int py = rem / t.PW, px = rem - py * t.PW;
        ptab[pix] = ((unsigned)nbi << 20) | ((unsigned)py << 10) | (unsigned)px;
    }
}

// Stride-2 inputs: every patch row is kept with its even and its odd columns de-interleaved ([even columns | odd columns]).
// The pixels one MFMA fragment gathers — every other column — are then neighbours in LDS, and the slot swizzle that is
// conflict-free for stride-1 layers is conflict-free for them as well (tools/lds_sim.py; with the plain row-major patch every B
// read of a stride-2 layer was a 2-way conflict: 21 % / 28 % of the LDS cycles of the forward / weight-gradient kernel).  A tap
// offset stays additive: column 2 px + dx lands at px + [(dx & 1) * half + (dx >> 1)].
__device__ __forceinline__ int cv_patch_col(int px, int PW, int is) { return is == 2 ? (px >> 1) + (px & 1) * ((PW + 1) >> 1) : px; }
// LDS pixel slot of patch pixel `pix` (row-major index, table entry e = nbi << 20 | py << 10 | px)
__device__ __forceinline__ int cv_patch_slot(int pix, unsigned e, const ConvTiling &t, int is) {
    return is == 2 ? ((int)(e >> 20) * t.PH + (int)((e >> 10) & 1023)) * t.PW + cv_patch_col((int)(e & 1023), t.PW, 2) : pix;
}

static int check_geom(const rick_conv_geom *g) {
    if (!g || g->N <= 0 || g->IH <= 0 || g->IW <= 0 || g->Ci <= 0 || g->OH <= 0 || g->OW <= 0 || g->Co <= 0 ||
        g->GH <= 0 || g->GW <= 0 || g->is <= 0 || g->os <= 0 || g->ntaps < 1 || g->ntaps > RICK_MAX_TAPS ||
        g->nslices < 1 || (g->split != 1 && g->split != 2))
        return RICK_EINVAL;
    for (int i = 0; i < g->ntaps; i++)
        if (g->wt[i] < 0 || g->wt[i] >= g->nslices) return RICK_EINVAL;
    if ((g->GH - 1) * g->os + g->oy0 >= g->OH || (g->GW - 1) * g->os + g->ox0 >= g->OW) return RICK_EINVAL;
    // 32-bit element offsets are used inside a tile
    if ((int64_t)g->N * g->IH * g->IW * g->Ci >= (1LL << 31) || (int64_t)g->N * g->OH * g->OW * g->Co >= (1LL << 31))
        return RICK_EINVAL;
    return 0;
}
