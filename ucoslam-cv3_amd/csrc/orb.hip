// Multi-scale ORB extractor for MI355X (gfx950): the per-frame work of
//   ucoslam::ORBextractor::detectAndCompute_impl / compute   src/featureextractors/ORBextractor.cpp:1139, :1247-1353
// behind the C ABI of include/ucoslam_hip.h.  Everything after the input upload runs on the GPU; nothing is
// read back until the keypoints/descriptors are final.
//
// Stage map (reference line -> kernel):
//   GaussianBlur 7x7 sigma 2 of the input (:1261-1263)                      -> blur7_kernel        (fixed point, LDS tile)
//   ComputePyramid: cubic resize chain level l-1 -> l (:1355-1393)          -> resize_cubic_kernel (11-bit taps from host tables)
//   per-cell cv::FAST(thr 20 | 7, nonmax) (:899-987)                        -> fast_score_kernel   (threshold-free strength map)
//                                                                              + cell_nms_kernel   (in-cell strict 3x3 maxima, raster-ordered compaction)
//   quota redistribution + retainBest per cell and per level (:994-1073)    -> select_kernel       (libstdc++ introselect data movement, see introselect.hpp)
//   IC_Angle (:79-106), rBRIEF (:113-153), border filter (:1120-1130),
//   coordinate rescale (:1228-1229), level-order concatenation (:1278-1301) -> describe_kernel     (one wave per keypoint)
//
// Two observations keep this exact AND simple:
//  (1) cornerScore<16> does not depend on the detection threshold once the pixel is a corner, and "corner at t" <=> score >= t.
//      A neighbour that is not a corner at t scores < t <= own score, so zeroing it (what cv::FAST's row buffers do) cannot
//      change the strict-maximum test.  Hence NMS is threshold-independent: a cell's keypoints at threshold t are its in-cell
//      strict local maxima with score >= t, and the 20 -> 7 fallback is a filter on one candidate list.
//  (2) the reference's 19-px reflected border around each level is never read (cells span [16,dim-16), the orientation disc
//      and the rotated pattern stay >= 1 px inside), so levels are stored un-padded.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <string>

#include "common.hpp"
#include "devframe.hpp"
#include "glibc_sincosf.hpp"
#include "introselect.hpp"
#include "../../include/ucoslam_hip_orb_pattern.inc"

namespace {

constexpr int kMaxLevels = 16;
constexpr int kMaxDim = 4095;          // candidate words pack x and y in 12 bits each
constexpr int kMaxCellsPerLevel = 2048;
constexpr int kNmsTileBytes = 16384;     // cell_nms: LDS staging of one cell's strength values
constexpr int kNmsPatchBytes = 20480;    // cell_nms (fused with the FAST strength): the cell's pixels + a 3-pixel frame
constexpr int kSelThreads = 1024;       // selection workgroup: 16 waves share one level's cells
constexpr int kSelWaves = kSelThreads / 64;
constexpr int PATCH_SIZE = 31, EDGE = 19;

struct LevelDesc {
    int w, h, pitch;
    int img_off;        // byte offset of the level inside one frame's pyramid (and score) buffer
    int nDesired;
    int nCells, cell_begin, cellCap, nfeaturesCell;
    int cand_off;       // entry offset of the level's candidate area inside one frame's candidate buffer
    int sel_off;        // entry offset of the level's selected list inside one frame's selection buffer
    int work_cap;       // entries this level may need in the selection workspace (= nCells*cellCap)
    int tiles_x, tiles_y, tile_begin;   // 64x16 tiles over the whole level
    int xtap_off, ytap_off;             // offsets into the tap tables (level l reads level l-1)
    int scaledPatchSize;
    float scale;
};

struct CellDesc {       // FAST-scanned interior of one cell, level coordinates, [x0,x1) x [y0,y1)
    short x0, y0, x1, y1;
    int skipped;        // the reference's `continue` (hX<=0 / hY<=0): cell never counted in the first pass
};

struct Plan {           // uploaded once per (size, params)
    int nlevels;
    int iniTh, minTh;
    int maxFeatures;
    int total_tiles, total_cells;
    int lvl_end;        // levels >= lvl_end are not needed by this instance's shard: the resize chain stops before them
    int sel_wg_begin[kMaxLevels + 1];   // select_kernel: level l is served by workgroups [sel_wg_begin[l], sel_wg_begin[l+1]) of a frame
    LevelDesc lv[kMaxLevels];
};

__device__ const signed char d_pattern[1024] = {UH_ORB_PATTERN_VALUES};
// the 749 pixels of the radius-15 disc (row half-widths umax = 15 15 15 15 14 14 14 13 13 12 11 10 9 8 6 3, ORBextractor.cpp:436-451) as (v << 8 | u & 255), padded to 12 x 64 with the centre (u = v = 0 adds nothing to either moment):
// describe_kernel's lanes take twelve pixels each, all loads in flight together, instead of 31 lanes walking a row each
struct DiscTable { short uv[768]; };
constexpr DiscTable make_disc_table() {
    constexpr int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    DiscTable t{};
    int n = 0;
    for (int v = -15; v <= 15; v++) {
        const int d = umax[v < 0 ? -v : v];
        for (int u = -d; u <= d; u++) t.uv[n++] = (short)(v * 256 + (u & 255));
    }
    for (; n < 768; n++) t.uv[n] = 0;
    return t;
}
__device__ const DiscTable d_disc = make_disc_table();

__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// ------------------------------------------------------------------------------------------------ blur
// cv::GaussianBlur(8U, 7x7, sigma=2, BORDER_REFLECT_101), OpenCV >= 4.3 fixed-point path: taps {18,34,48,56,48,34,18}/256,
// horizontal 8.8 accumulate, vertical 16.16 accumulate, (v + 2^15) >> 16.  Tile 64 x 16, halo 3.
// Four pixels per thread and pass: a row of the tile + halo is staged as dwords (LDS column c <-> image column tx0 - 4 + c; tiles whose
// halo lies inside the image — four of five — fetch it as 19 dword loads per row, the border tiles byte by byte with the reflection),
// the horizontal pass reads three dwords and forms the four 7-tap sums with v_alignbyte + v_dot4_u32_u8 (taps fit a byte, sums 16 bits),
// the vertical pass reads 7 x 4 sixteen-bit sums as 64-bit words and writes its four pixels as one dword.  Same integers as the
// one-pixel-per-thread form it replaces (~300 -> ~110 instructions per thread and tile).
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
__global__ __launch_bounds__(256) void blur7_kernel(const uint8_t* __restrict__ src, int w, int h, size_t src_stride,
                                                    size_t src_frame_stride, uint8_t* __restrict__ dst, int dst_pitch,
                                                    size_t dst_frame_stride) {
    constexpr int TW = 64, TH = 16, R = 3, ROWS = TH + 2 * R, SW = 76;   // SW: 19 dwords = columns tx0 - 4 .. tx0 + 71
    __shared__ __attribute__((aligned(16))) uint8_t s_in[ROWS][SW];
    __shared__ __attribute__((aligned(16))) uint16_t s_h[ROWS][TW];
    const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    const uint8_t* in = src + (size_t)blockIdx.z * src_frame_stride;
    uint8_t* out = dst + (size_t)blockIdx.z * dst_frame_stride;
    const bool inside = tx0 >= 4 && tx0 + 72 <= w && ty0 >= R && ty0 + TH + R <= h;   // (uniform in the workgroup)
    if (inside) {
        const uint8_t* base = in + (size_t)(ty0 - R) * src_stride + (tx0 - 4);
        for (int i = threadIdx.x; i < ROWS * (SW / 4); i += 256) {
            const int ly = i / (SW / 4), c4 = i - ly * (SW / 4);
            reinterpret_cast<uint32_t*>(&s_in[ly][0])[c4] = *reinterpret_cast<const u32_unaligned*>(base + (size_t)ly * src_stride + 4 * c4);
        }
    } else {
        for (int i = threadIdx.x; i < ROWS * (TW + 2 * R); i += 256) {
            const int ly = i / (TW + 2 * R), lx = i - ly * (TW + 2 * R);
            const int gy = reflect101(ty0 + ly - R, h), gx = reflect101(tx0 + lx - R, w);
            s_in[ly][lx + 1] = in[(size_t)gy * src_stride + gx];
        }
    }
    __syncthreads();
    constexpr uint32_t W0 = 18u | (34u << 8) | (48u << 16) | (56u << 24), W1 = 48u | (34u << 8) | (18u << 16);
    for (int i = threadIdx.x; i < ROWS * (TW / 4); i += 256) {
        const int ly = i / (TW / 4), cg = i - ly * (TW / 4);
        const uint32_t* row = reinterpret_cast<const uint32_t*>(&s_in[ly][0]) + cg;
        const uint32_t d0 = row[0], d1 = row[1], d2 = row[2];   // columns 4 cg .. 4 cg + 11; output j's window starts at column 4 cg + j + 1
        const uint32_t s0 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 1), W0, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 1), W1, 0u, false), false);
        const uint32_t s1 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 2), W0, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 2), W1, 0u, false), false);
        const uint32_t s2 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 3), W0, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 3), W1, 0u, false), false);
        const uint32_t s3 = __builtin_amdgcn_udot4(d1, W0, __builtin_amdgcn_udot4(d2, W1, 0u, false), false);
        *reinterpret_cast<uint2*>(&s_h[ly][4 * cg]) = make_uint2(s0 | (s1 << 16), s2 | (s3 << 16));
    }
    __syncthreads();
    {
        const int ly = threadIdx.x >> 4, cg = threadIdx.x & 15;
        const int gx = tx0 + 4 * cg, gy = ty0 + ly;
        if (gx < w && gy < h) {
            uint2 r[7];
#pragma unroll
            for (int k = 0; k < 7; k++) r[k] = *reinterpret_cast<const uint2*>(&s_h[ly + k][4 * cg]);
            auto col = [&](int j, int k) -> uint32_t { const uint32_t v = (j & 2) ? r[k].y : r[k].x; return (j & 1) ? (v >> 16) : (v & 0xFFFFu); };
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t sv = 18u * (col(j, 0) + col(j, 6)) + 34u * (col(j, 1) + col(j, 5)) + 48u * (col(j, 2) + col(j, 4)) + 56u * col(j, 3);
                o[j] = (sv + 32768u) >> 16;
            }
            uint8_t* op = out + (size_t)gy * dst_pitch + gx;
            if (gx + 3 < w) *reinterpret_cast<u32_unaligned*>(op) = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
            else {
#pragma unroll
                for (int j = 0; j < 4; j++) if (gx + j < w) op[j] = (uint8_t)o[j];
            }
        }
    }
}

// A frame that lies in pinned host memory comes over with 16-byte loads (one PCIe read request per 64 bytes of a wave's row segment):
// blur7_kernel reading it in place took 49 us per 1241 x 376 frame (byte loads, 1.5x halo re-reads over the link), this copy + the
// blur from HBM take 15.  Rows of `w` bytes, source row stride `stride`, packed destination.
__global__ __launch_bounds__(256) void ingest_kernel(const uint8_t* __restrict__ src, int w, int h, size_t stride, uint8_t* __restrict__ dst) {
    if (stride == (size_t)w) {   // one contiguous array: 16-byte chunks from the (16-byte aligned) base
        const size_t total = (size_t)w * h, i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
        if (i + 16 <= total) *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(src + i);
        else for (size_t k = i; k < total; k++) dst[k] = src[k];
        return;
    }
    const int chunks = (w + 15) / 16;
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= chunks * h) return;
    const int y = id / chunks, x = (id - y * chunks) * 16;
    const uint8_t* sp = src + (size_t)y * stride + x;
    uint8_t* dp = dst + (size_t)y * w + x;
    if (x + 16 <= w) {
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = *reinterpret_cast<const u32_unaligned*>(sp + 4 * k);
#pragma unroll
        for (int k = 0; k < 4; k++) *reinterpret_cast<u32_unaligned*>(dp + 4 * k) = v[k];
    } else for (int k = 0; x + k < w; k++) dp[k] = sp[k];
}

// An interleaved BGR (3 bytes) or BGRA (4 bytes) frame to gray on its way in: cv::cvtColor(in, gray, COLOR_BGR2GRAY) as FrameExtractor does
// for three-channel input (src/utils/frameextractor.cpp:2960,3046) — OpenCV's 8-bit form: (B * 3735 + G * 19235 + R * 9798 + 2^14) >> 15
// (RGB2Gray<uchar>, 15-bit coefficients: OpenCV >= 3.4.2 / 4.x — the same line as the >= 4.3 GaussianBlur / resize semantics this extractor
// restates; OpenCV 3.0 - 3.4.1 used 14-bit coefficients 1868 / 9617 / 4899, which differ by one grey level on many pixels: a host built
// against those must convert on its side and hand over gray frames.  Unpinned like the rest of the extractor, oracle_bgr2gray restates it).  A thread
// converts four pixels (12 or 16 source bytes, dword loads: the source may be pinned host memory) into one dword of the packed gray frame.
__global__ __launch_bounds__(256) void ingest_bgr_kernel(const uint8_t* __restrict__ src, int w, int h, size_t stride, int cn, uint8_t* __restrict__ dst) {
    const int quads = (w + 3) / 4;
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= quads * h) return;
    const int y = id / quads, x = (id - y * quads) * 4;
    const uint8_t* sp = src + (size_t)y * stride + (size_t)x * cn;
    uint8_t px[16];
    const int npx = min(4, w - x), nbytes = npx * cn;
    if (npx == 4 && ((reinterpret_cast<uintptr_t>(sp) & 3) == 0)) {
#pragma unroll
        for (int k = 0; k < 4; k++) if (k * 4 < nbytes) *reinterpret_cast<uint32_t*>(px + 4 * k) = *reinterpret_cast<const uint32_t*>(sp + 4 * k);
    } else {
        for (int k = 0; k < nbytes; k++) px[k] = sp[k];
    }
    uint8_t g[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (k < npx) g[k] = (uint8_t)(((int)px[k * cn] * 3735 + (int)px[k * cn + 1] * 19235 + (int)px[k * cn + 2] * 9798 + (1 << 14)) >> 15);
    uint8_t* dp = dst + (size_t)y * w + x;
    if (npx == 4 && (reinterpret_cast<uintptr_t>(dp) & 3) == 0) *reinterpret_cast<uint32_t*>(dp) = (uint32_t)g[0] | ((uint32_t)g[1] << 8) | ((uint32_t)g[2] << 16) | ((uint32_t)g[3] << 24);
    else for (int k = 0; k < npx; k++) dp[k] = g[k];
}

// plain copy of the input into level 0 (doGaussianBlur == false)
__global__ void copy_kernel(const uint8_t* __restrict__ src, int w, int h, size_t src_stride, size_t src_frame_stride,
                            uint8_t* __restrict__ dst, int dst_pitch, size_t dst_frame_stride) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x < w) dst[(size_t)blockIdx.z * dst_frame_stride + (size_t)y * dst_pitch + x] =
        src[(size_t)blockIdx.z * src_frame_stride + (size_t)y * src_stride + x];
}

// i / d for i < 2^16 <= 2^32 / d with the reciprocal worked out once per workgroup (m = ceil(2^32 / d); d = 1: m wraps to 0): one
// v_mul_hi_u32 instead of the ~25-instruction division sequence, which the cell kernel ran a dozen times per thread.
__device__ __forceinline__ unsigned div_magic(int d) { return (unsigned)(0xFFFFFFFFu / (unsigned)d) + 1u; }
__device__ __forceinline__ int div_by(int i, int d, unsigned m) { return d == 1 ? i : (int)__umulhi((unsigned)i, m); }

// ------------------------------------------------------------------------------------------------ pyramid
// cv::resize(INTER_CUBIC) 8UC1 reference path: int32 horizontal pass with 11-bit taps, int32 vertical pass,
// (v + 2^21) >> 22, saturate.  Taps clamp at the source ROI edge.  ofs = floor(src coord), 4 shorts per output index.
// Tile = 64 x 16 output pixels.  The source footprint of the tile (<= 144 x 40 for scale factors up to ~2) is staged in
// LDS with coalesced row loads, the horizontal pass writes int32 rows to LDS once per SOURCE row (shared by the ~1.2
// output rows that use it), the vertical pass reads four of them per output pixel: ~14 LDS reads per pixel instead of 16
// uncoalesced byte loads from L1/L2.  Arithmetic is exactly the fixed-point reference path (see the oracle).
template <int HW>
struct ResizeLds {   // staging of one 64 x 16 output tile: source footprint (dword-staged from a 4-aligned column) + horizontal pass (HW columns)
    alignas(16) uint8_t src[40][148];
    alignas(16) int h[40][HW];
};
struct ResizePairLds : ResizeLds<96> {   // the pair kernel adds the rectangle of the intermediate level
    uint8_t mid[40][96];
};
template <typename LdsT>
__device__ __forceinline__ void resize_cubic_tile(LdsT& L, const uint8_t* __restrict__ S, int sw, int sh, int spitch,
                                                  uint8_t* __restrict__ D, int dw, int dh, int dpitch, const int* __restrict__ xofs,
                                                  const short* __restrict__ xcoef, const int* __restrict__ yofs,
                                                  const short* __restrict__ ycoef, int tx0, int ty0) {
    constexpr int TW = 64, TH = 16, SWMAX = 144, SHMAX = 40;
    const int xl = min(tx0 + TW, dw) - 1, yl = min(ty0 + TH, dh) - 1;
    const int rx0 = min(max(xofs[tx0] - 1, 0), sw - 1), rx1 = min(max(xofs[xl] + 2, 0), sw - 1);
    const int ry0 = min(max(yofs[ty0] - 1, 0), sh - 1), ry1 = min(max(yofs[yl] + 2, 0), sh - 1);
    const int srcw = rx1 - rx0 + 1, srch = ry1 - ry0 + 1;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int gx = tx0 + lx;
    if (srcw > SWMAX || srch > SHMAX) {   // extreme scale factors: direct evaluation, one thread per output pixel
        for (int j = 0; j < 4; j++) {
            const int gy = ty0 + ly + 4 * j;
            if (gx >= dw || gy >= dh) continue;
            const int xo = xofs[gx], yo = yofs[gy];
            int acc = 0;
            for (int k = 0; k < 4; k++) {
                const uint8_t* row = S + (size_t)min(max(yo - 1 + k, 0), sh - 1) * spitch;
                int hsum = 0;
                for (int t = 0; t < 4; t++) hsum += row[min(max(xo - 1 + t, 0), sw - 1)] * xcoef[gx * 4 + t];
                acc += hsum * ycoef[gy * 4 + k];
            }
            const int v = (acc + (1 << 21)) >> 22;
            D[(size_t)gy * dpitch + gx] = (uint8_t)min(max(v, 0), 255);
        }
        return;
    }
    // the footprint as dwords from a 4-aligned source column (a few columns more than needed; what lies past the level's last column is
    // never a tap): ~2 loads per thread instead of ~12 single bytes
    const int rx0a = rx0 & ~3, ndw = ((rx1 - rx0a) >> 2) + 1;
    for (int i = threadIdx.x; i < srch * ndw; i += 256) {
        const int r = i / ndw, c4 = i - r * ndw;
        reinterpret_cast<uint32_t*>(&L.src[r][0])[c4] = *reinterpret_cast<const u32_unaligned*>(S + (size_t)(ry0 + r) * spitch + rx0a + 4 * c4);
    }
    __syncthreads();
    if (gx < dw) {   // horizontal pass: this thread's column, every staged source row
        const int xo = xofs[gx];
        int c[4], sx[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { c[k] = xcoef[gx * 4 + k]; sx[k] = min(max(xo - 1 + k, 0), sw - 1) - rx0a; }
        for (int r = ly; r < srch; r += 4)
            L.h[r][lx] = L.src[r][sx[0]] * c[0] + L.src[r][sx[1]] * c[1] + L.src[r][sx[2]] * c[2] + L.src[r][sx[3]] * c[3];
    }
    __syncthreads();
    {   // vertical pass: four neighbouring pixels of one row per thread (the row's four taps are read once, the pixels leave as one dword)
        const int ry = threadIdx.x >> 4, cg = threadIdx.x & 15;
        const int gy = ty0 + ry, gx0 = tx0 + 4 * cg;
        if (gy < dh && gx0 < dw) {
            const int yo = yofs[gy];
            int acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int4 hv = *reinterpret_cast<const int4*>(&L.h[min(max(yo - 1 + k, 0), sh - 1) - ry0][4 * cg]);
                const int cy = ycoef[gy * 4 + k];
                acc[0] += hv.x * cy; acc[1] += hv.y * cy; acc[2] += hv.z * cy; acc[3] += hv.w * cy;
            }
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) o[j] = (uint32_t)min(max((acc[j] + (1 << 21)) >> 22, 0), 255);
            uint8_t* op = D + (size_t)gy * dpitch + gx0;
            if (gx0 + 3 < dw) *reinterpret_cast<u32_unaligned*>(op) = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
            else {
#pragma unroll
                for (int j = 0; j < 4; j++) if (gx0 + j < dw) op[j] = (uint8_t)o[j];
            }
        }
    }
}
__global__ __launch_bounds__(256) void resize_cubic_kernel(const uint8_t* __restrict__ src, int sw, int sh, int spitch,
                                                           uint8_t* __restrict__ dst, int dw, int dh, int dpitch,
                                                           size_t frame_stride, const int* __restrict__ xofs,
                                                           const short* __restrict__ xcoef, const int* __restrict__ yofs,
                                                           const short* __restrict__ ycoef) {
    __shared__ ResizeLds<64> L;
    resize_cubic_tile(L, src + (size_t)blockIdx.z * frame_stride, sw, sh, spitch, dst + (size_t)blockIdx.z * frame_stride, dw, dh, dpitch,
                      xofs, xcoef, yofs, ycoef, blockIdx.x * 64, blockIdx.y * 16);
}

// TWO levels per launch: levels a and a+1 from level a-1.  A dependent launch costs ~9.5 us on MI355X whatever it computes (a HIP graph does
// not change that), and the seven resizes of the pyramid were 66 of the extractor's 186 us per 4-frame launch set.  Workgroups
// [0, tilesA) compute the tiles of level a as before; workgroups [tilesA, tilesA + tilesB) compute a tile of level a+1 WITHOUT waiting for
// level a: they rebuild the rectangle of level a their tile reads (<= 96 x 40 pixels) in LDS from level a-1 — the same taps, the same
// fixed-point arithmetic, hence the same bytes the other workgroups write — and resize that.  Level a is computed 2.2 times, level a+1 once;
// the chain is four launches instead of seven.  Measured (1241x376, 8 levels): one frame 123 -> 112 us per extraction (8 launches), but
// 187 -> 193 us for four frames and 266 -> 290 us for eight — the recomputation outweighs three saved launches as soon as the resizes have
// real work — so the host takes this path for batches of at most two frames (the single-camera tracking case).  (Two earlier fusions
// are in DESIGN.md: four levels per launch recomputed 3.5 times the pyramid; one cooperative launch with barriers between levels lost to
// the L2-bypassing loads it needs.)
struct LevelTaps { const int* xofs; const short* xcoef; const int* yofs; const short* ycoef; };
__global__ __launch_bounds__(256) void resize_pair_kernel(const uint8_t* __restrict__ pyr, size_t frame_stride, LevelDesc S0, LevelDesc LA, LevelDesc LB,
                                                          LevelTaps ta, LevelTaps tb, int tilesA) {
    __shared__ ResizePairLds L;
    const uint8_t* F = pyr + (size_t)blockIdx.z * frame_stride;
    uint8_t* Fw = const_cast<uint8_t*>(F);
    if ((int)blockIdx.x < tilesA) {
        const int tx = (LA.w + 63) / 64, t = blockIdx.x, ty = t / tx;
        resize_cubic_tile(L, F + S0.img_off, S0.w, S0.h, S0.pitch, Fw + LA.img_off, LA.w, LA.h, LA.pitch, ta.xofs, ta.xcoef, ta.yofs, ta.ycoef,
                          (t - ty * tx) * 64, ty * 16);
        return;
    }
    constexpr int TW = 64, TH = 16;
    const int txb = (LB.w + 63) / 64, t = blockIdx.x - tilesA, tyb = t / txb;
    const int tx0 = (t - tyb * txb) * TW, ty0 = tyb * TH;
    const int xl = min(tx0 + TW, LB.w) - 1, yl = min(ty0 + TH, LB.h) - 1;
    // the rectangle of level a this tile reads, and the rectangle of level a-1 THAT reads (the host checked that both fit the staging area)
    const int ax0 = min(max(tb.xofs[tx0] - 1, 0), LA.w - 1), ax1 = min(max(tb.xofs[xl] + 2, 0), LA.w - 1);
    const int ay0 = min(max(tb.yofs[ty0] - 1, 0), LA.h - 1), ay1 = min(max(tb.yofs[yl] + 2, 0), LA.h - 1);
    const int aw = ax1 - ax0 + 1, ah = ay1 - ay0 + 1;
    const int sx0 = min(max(ta.xofs[ax0] - 1, 0), S0.w - 1), sx1 = min(max(ta.xofs[ax1] + 2, 0), S0.w - 1);
    const int sy0 = min(max(ta.yofs[ay0] - 1, 0), S0.h - 1), sy1 = min(max(ta.yofs[ay1] + 2, 0), S0.h - 1);
    const int sw0 = sx1 - sx0 + 1, sh0 = sy1 - sy0 + 1;
    const uint8_t* S = F + S0.img_off;
    (void)sw0;
    const int sx0a = sx0 & ~3, ndw = ((sx1 - sx0a) >> 2) + 1;   // (dword staging from a 4-aligned column, as in resize_cubic_tile)
    const unsigned m_ndw = div_magic(ndw), m_aw = div_magic(aw);
    for (int i = threadIdx.x; i < sh0 * ndw; i += 256) {
        const int r = div_by(i, ndw, m_ndw), c4 = i - r * ndw;
        reinterpret_cast<uint32_t*>(&L.src[r][0])[c4] = *reinterpret_cast<const u32_unaligned*>(S + (size_t)(sy0 + r) * S0.pitch + sx0a + 4 * c4);
    }
    __syncthreads();
    // level a, horizontal: column ax0 + c of every staged source row
    for (int i = threadIdx.x; i < aw * sh0; i += 256) {
        const int r = div_by(i, aw, m_aw), c = i - r * aw, gx = ax0 + c;
        const int xo = ta.xofs[gx];
        int acc = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) acc += L.src[r][min(max(xo - 1 + k, 0), S0.w - 1) - sx0a] * ta.xcoef[gx * 4 + k];
        L.h[r][c] = acc;
    }
    __syncthreads();
    // level a, vertical -> the rectangle's bytes
    for (int i = threadIdx.x; i < aw * ah; i += 256) {
        const int r = div_by(i, aw, m_aw), c = i - r * aw, gy = ay0 + r;
        const int yo = ta.yofs[gy];
        int acc = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) acc += L.h[min(max(yo - 1 + k, 0), S0.h - 1) - sy0][c] * ta.ycoef[gy * 4 + k];
        L.mid[r][c] = (uint8_t)min(max((acc + (1 << 21)) >> 22, 0), 255);
    }
    __syncthreads();
    // level a+1 from the rectangle: horizontal (64 columns x ah rows), then vertical
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6, gx = tx0 + lx;
    if (gx < LB.w) {
        const int xo = tb.xofs[gx];
        int c[4], sx[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { c[k] = tb.xcoef[gx * 4 + k]; sx[k] = min(max(xo - 1 + k, 0), LA.w - 1) - ax0; }
        for (int r = ly; r < ah; r += 4)
            L.h[r][lx] = L.mid[r][sx[0]] * c[0] + L.mid[r][sx[1]] * c[1] + L.mid[r][sx[2]] * c[2] + L.mid[r][sx[3]] * c[3];
    }
    __syncthreads();
    {   // vertical: four neighbouring pixels of one row per thread, one dword store
        const int ry = threadIdx.x >> 4, cg = threadIdx.x & 15;
        const int gy = ty0 + ry, gx0 = tx0 + 4 * cg;
        if (gy < LB.h && gx0 < LB.w) {
            const int yo = tb.yofs[gy];
            int acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int4 hv = *reinterpret_cast<const int4*>(&L.h[min(max(yo - 1 + k, 0), LA.h - 1) - ay0][4 * cg]);
                const int cy = tb.ycoef[gy * 4 + k];
                acc[0] += hv.x * cy; acc[1] += hv.y * cy; acc[2] += hv.z * cy; acc[3] += hv.w * cy;
            }
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) o[j] = (uint32_t)min(max((acc[j] + (1 << 21)) >> 22, 0), 255);
            uint8_t* op = Fw + LB.img_off + (size_t)gy * LB.pitch + gx0;
            if (gx0 + 3 < LB.w) *reinterpret_cast<u32_unaligned*>(op) = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
            else {
#pragma unroll
                for (int j = 0; j < 4; j++) if (gx0 + j < LB.w) op[j] = (uint8_t)o[j];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ the whole pyramid in ONE launch
// Blur + every resize of ComputePyramid (ORBextractor.cpp:1261-1263, 1355-1393) by spatial tiles: level l depends on level l - 1 only
// through a 4-tap footprint, so a workgroup that owns the same share R_l of EVERY level (tile (i, j) of an NTX x NTY partition) builds
// them all out of LDS — at level l the rectangle C_l = R_l + what its C_{l+1} reads (the host works the rectangles out top down from the
// tap tables: the halo accumulates to ~26 pixels at level 0), computed from the C_{l-1} it holds, only R_l going to HBM.  Four or five
// dependent launches of 5-10 us become one; halo pixels are computed by two or more workgroups (the same integers).  Arithmetic: the
// blur and resize paths above, restated on LDS rectangles (horizontal int32 sums per source row and destination column, vertical pass
// over bands of 32 destination rows).
constexpr int kPyrThreads = 1024, kPyrBand = 96;   // destination rows per vertical-pass band
struct PyrTile { short r[kMaxLevels][4], c[kMaxLevels][4]; };   // x0 y0 x1 y1 (exclusive): what the tile writes / computes of each level
struct PyrPlan { int ntiles, nlevels; int szA, szB, szRaw, szH16, szH32; };   // LDS areas in bytes
struct PyrTaps { const int* xofs; const short* xcoef; const int* yofs; const short* ycoef; };

__global__ __launch_bounds__(kPyrThreads) void pyramid_fused_kernel(const uint8_t* __restrict__ src, int w, int h, size_t src_stride, size_t src_frame_stride,
                                                                   uint8_t* __restrict__ pyr, size_t frame_stride, const Plan plan, const PyrPlan pp,
                                                                   const PyrTile* __restrict__ tiles, const PyrTaps taps, int blur) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_pyr[];
    __shared__ short s_yrow[kPyrBand][4], s_ycf[kPyrBand][4];
    const PyrTile& T = tiles[blockIdx.x];
    const uint8_t* in = src + (size_t)blockIdx.z * src_frame_stride;
    uint8_t* F = pyr + (size_t)blockIdx.z * frame_stride;
    uint8_t* bufA = s_pyr;                       // levels 0, 2, 4, ..
    uint8_t* bufB = s_pyr + pp.szA;              // levels 1, 3, 5, .. (level 0's raw tile and 16-bit sums live here first)
    int* h32 = reinterpret_cast<int*>(s_pyr + pp.szA + (pp.szB > pp.szRaw + pp.szH16 ? pp.szB : pp.szRaw + pp.szH16));
    const int tid = threadIdx.x;
    // ---- level 0: the blurred (or copied) rectangle C_0
    {
        const int cx0 = T.c[0][0], cy0 = T.c[0][1], cw = T.c[0][2] - cx0, ch = T.c[0][3] - cy0;
        const int rx0 = T.r[0][0], ry0 = T.r[0][1], rx1 = T.r[0][2], ry1 = T.r[0][3];
        const LevelDesc& L0 = plan.lv[0];
        if (cw > 0 && ch > 0) {
            const int pa = (cw + 3) & ~3;
            if (blur) {
                uint8_t* raw = bufB;
                uint16_t* h16 = reinterpret_cast<uint16_t*>(bufB + pp.szRaw);
                const int rw = cw + 6, rh = ch + 6, rp = (rw + 3) & ~3;
                {   // a thread keeps its column: the reflected source column is worked out once
                    const int rstep = max(kPyrThreads / rw, 1), lx = tid % rw, r0 = tid / rw;
                    if (r0 < rstep) {
                        const int gx = reflect101(cx0 + lx - 3, w);
                        for (int ly = r0; ly < rh; ly += 8 * rstep) {   // eight loads in flight (a dependent global load a row would be ~1 us each)
                            uint8_t v[8];
#pragma unroll
                            for (int u = 0; u < 8; u++) { const int y = ly + u * rstep; v[u] = y < rh ? in[(size_t)reflect101(cy0 + y - 3, h) * src_stride + gx] : (uint8_t)0; }
#pragma unroll
                            for (int u = 0; u < 8; u++) { const int y = ly + u * rstep; if (y < rh) raw[y * rp + lx] = v[u]; }
                        }
                    }
                }
                __syncthreads();
                const int rstep = max(kPyrThreads / cw, 1), lx = tid % cw, r0 = tid / cw;
                if (r0 < rstep)
                    for (int ly = r0; ly < rh; ly += rstep) {   // horizontal 8.8 sums
                        const uint8_t* q = raw + ly * rp + lx;
                        h16[ly * pa + lx] = (uint16_t)(18u * (q[0] + q[6]) + 34u * (q[1] + q[5]) + 48u * (q[2] + q[4]) + 56u * q[3]);
                    }
                __syncthreads();
                if (r0 < rstep) {
                    const int gx = cx0 + lx;
                    const bool xin = gx >= rx0 && gx < rx1;
                    for (int ly = r0; ly < ch; ly += rstep) {   // vertical 16.16 sums
                        const uint16_t* q = h16 + ly * pa + lx;
                        const uint32_t sv = 18u * ((uint32_t)q[0] + q[6 * pa]) + 34u * ((uint32_t)q[pa] + q[5 * pa]) + 48u * ((uint32_t)q[2 * pa] + q[4 * pa]) + 56u * q[3 * pa];
                        const uint8_t o = (uint8_t)((sv + 32768u) >> 16);
                        bufA[ly * pa + lx] = o;
                        const int gy = cy0 + ly;
                        if (xin && gy >= ry0 && gy < ry1) F[L0.img_off + (size_t)gy * L0.pitch + gx] = o;
                    }
                }
            } else {
                const unsigned m_cw = div_magic(cw);
                for (int i = tid; i < cw * ch; i += kPyrThreads) {
                    const int ly = div_by(i, cw, m_cw), lx = i - ly * cw;
                    const int gx = cx0 + lx, gy = cy0 + ly;
                    const uint8_t o = in[(size_t)gy * src_stride + gx];
                    bufA[ly * pa + lx] = o;
                    if (gx >= rx0 && gx < rx1 && gy >= ry0 && gy < ry1) F[L0.img_off + (size_t)gy * L0.pitch + gx] = o;
                }
            }
        }
        __syncthreads();
    }
    // ---- levels 1 ..: C_l from C_{l-1}
    for (int l = 1; l < pp.nlevels; l++) {
        const LevelDesc& S = plan.lv[l - 1];
        const LevelDesc& D = plan.lv[l];
        const uint8_t* P = (l & 1) ? bufA : bufB;
        uint8_t* Q = (l & 1) ? bufB : bufA;
        const int px0 = T.c[l - 1][0], py0 = T.c[l - 1][1], pw = T.c[l - 1][2] - px0;
        const int ppitch = (pw + 3) & ~3;
        const int cx0 = T.c[l][0], cy0 = T.c[l][1], cw = T.c[l][2] - cx0, ch = T.c[l][3] - cy0;
        const int rx0 = T.r[l][0], ry0 = T.r[l][1], rx1 = T.r[l][2], ry1 = T.r[l][3];
        if (cw <= 0 || ch <= 0) continue;     // (uniform; the tiles above this level are empty too)
        const int qp = (cw + 3) & ~3;
        const int* xofs = taps.xofs + D.xtap_off; const short* xcoef = taps.xcoef + (size_t)D.xtap_off * 4;
        const int* yofs = taps.yofs + D.ytap_off; const short* ycoef = taps.ycoef + (size_t)D.ytap_off * 4;
        // a thread keeps its destination column through both passes: its four source columns and horizontal taps are worked out once
        const int rstep = max(kPyrThreads / cw, 1), c = tid % cw, r0 = tid / cw, gx = cx0 + c;
        const bool mine = r0 < rstep;
        int sx[4], cxk[4];
        {
            const int xo = xofs[gx];
#pragma unroll
            for (int k = 0; k < 4; k++) { sx[k] = min(max(xo - 1 + k, 0), S.w - 1) - px0; cxk[k] = xcoef[gx * 4 + k]; }
        }
        const bool xin = gx >= rx0 && gx < rx1;
        for (int by = 0; by < ch; by += kPyrBand) {
            const int bh = min(kPyrBand, ch - by);
            const int gy_first = cy0 + by, gy_last = gy_first + bh - 1;
            const int sy0 = min(max(yofs[gy_first] - 1, 0), S.h - 1), sy1 = min(max(yofs[gy_last] + 2, 0), S.h - 1);
            const int srows = sy1 - sy0 + 1;
            if (tid < bh) {   // the band's vertical taps, once (a global load per row inside the pass below would be a dependent ~1 us each)
                const int gy = gy_first + tid, yo = yofs[gy];
                const short* yc = ycoef + gy * 4;
#pragma unroll
                for (int k = 0; k < 4; k++) { s_yrow[tid][k] = (short)(min(max(yo - 1 + k, 0), S.h - 1) - sy0); s_ycf[tid][k] = yc[k]; }
            }
            if (mine)
                for (int r = r0; r < srows; r += rstep) {   // horizontal pass: every source row of the band
                    const uint8_t* row = P + (sy0 + r - py0) * ppitch;
                    h32[r * qp + c] = row[sx[0]] * cxk[0] + row[sx[1]] * cxk[1] + row[sx[2]] * cxk[2] + row[sx[3]] * cxk[3];
                }
            __syncthreads();
            if (mine)
                for (int r = r0; r < bh; r += rstep) {      // vertical pass
                    const int gy = gy_first + r;
                    int acc = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) acc += h32[s_yrow[r][k] * qp + c] * s_ycf[r][k];
                    const uint8_t o = (uint8_t)min(max((acc + (1 << 21)) >> 22, 0), 255);
                    Q[(by + r) * qp + c] = o;
                    if (xin && gy >= ry0 && gy < ry1) F[D.img_off + (size_t)gy * D.pitch + gx] = o;
                }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------ FAST strength map
// score(p) = max over the 16 arcs of 9 contiguous circle pixels of min(|I_k - I_p| signed the same way) - 1, clamped to >= 0:
// the value cv::cornerScore<16> returns for any threshold <= score.  Pixels closer than 3 to the level edge score 0.
__device__ __forceinline__ int fast_strength(const int (&d)[25]) {
    int best = 0;   // a0 starts at threshold 0
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        int a = min(min(min(d[k + 1], d[k + 2]), min(d[k + 3], d[k + 4])), min(min(d[k + 5], d[k + 6]), min(d[k + 7], d[k + 8])));
        best = max(best, max(min(a, d[k]), min(a, d[k + 9])));
        int b = max(max(max(d[k + 1], d[k + 2]), max(d[k + 3], d[k + 4])), max(max(d[k + 5], d[k + 6]), max(d[k + 7], d[k + 8])));
        best = max(best, -min(max(b, d[k]), max(b, d[k + 9])));
    }
    return best - 1;
}

__global__ __launch_bounds__(256) void fast_score_kernel(const Plan plan, const uint8_t* __restrict__ pyr,
                                                         uint8_t* __restrict__ score, size_t frame_stride) {
    constexpr int TW = 64, TH = 16, R = 3;
    __shared__ uint8_t s_t[TH + 2 * R][TW + 2 * R + 2];
    int lvl = 0;
    const int tile = blockIdx.x;
    while (lvl + 1 < plan.nlevels && tile >= plan.lv[lvl + 1].tile_begin) ++lvl;
    const LevelDesc& L = plan.lv[lvl];
    const int t = tile - L.tile_begin;
    const int ty0 = (t / L.tiles_x) * TH, tx0 = (t % L.tiles_x) * TW;
    const uint8_t* img = pyr + (size_t)blockIdx.y * frame_stride + L.img_off;
    uint8_t* out = score + (size_t)blockIdx.y * frame_stride + L.img_off;
    for (int i = threadIdx.x; i < (TH + 2 * R) * (TW + 2 * R); i += 256) {
        int ly = i / (TW + 2 * R), lx = i - ly * (TW + 2 * R);
        int gy = min(max(ty0 + ly - R, 0), L.h - 1), gx = min(max(tx0 + lx - R, 0), L.w - 1);
        s_t[ly][lx] = img[(size_t)gy * L.pitch + gx];
    }
    __syncthreads();
    const int lx = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int ly = (threadIdx.x >> 6) + 4 * r;
        const int gx = tx0 + lx, gy = ty0 + ly;
        if (gx >= L.w || gy >= L.h) continue;
        int s = 0;
        if (gx >= 3 && gy >= 3 && gx < L.w - 3 && gy < L.h - 3) {
            const uint8_t* c = &s_t[ly + R][lx + R];
            const int v = c[0];
            int d[25];
            constexpr int stride = TW + 2 * R + 2;
            d[0] = v - c[3 * stride + 0];   d[1] = v - c[3 * stride + 1];   d[2] = v - c[2 * stride + 2];
            d[3] = v - c[1 * stride + 3];   d[4] = v - c[3];                d[5] = v - c[-1 * stride + 3];
            d[6] = v - c[-2 * stride + 2];  d[7] = v - c[-3 * stride + 1];  d[8] = v - c[-3 * stride];
            d[9] = v - c[-3 * stride - 1];  d[10] = v - c[-2 * stride - 2]; d[11] = v - c[-1 * stride - 3];
            d[12] = v - c[-3];              d[13] = v - c[1 * stride - 3];  d[14] = v - c[2 * stride - 2];
            d[15] = v - c[3 * stride - 1];
#pragma unroll
            for (int k = 16; k < 25; k++) d[k] = d[k - 16];
            s = max(fast_strength(d), 0);
        }
        out[(size_t)gy * L.pitch + gx] = (uint8_t)s;
    }
}

// ------------------------------------------------------------------------------------------------ per-cell NMS
// One workgroup per (frame, cell).  Candidates = in-cell strict 3x3 maxima with score >= minTh, written in raster order as
// (score<<24 | y<<12 | x) in level coordinates; n7 = their count, n20 = how many of them reach iniTh.
// FUSED: the strength values of the cell are computed here, from the pyramid level (the cell's pixels + a 3-pixel frame staged in LDS) —
// every pixel belongs to exactly one cell, so nothing is computed twice, and the strength map (one launch, one write and one read of
// every level) disappears from the extractor's chain.  Plans with a cell too large for the staging buffers keep the two-launch form.
template <bool FUSED>
__global__ __launch_bounds__(256) void cell_nms_kernel(const Plan plan, const CellDesc* __restrict__ cells,
                                                       const uint8_t* __restrict__ score /* FUSED: the pyramid */, size_t frame_stride,
                                                       uint32_t* __restrict__ cand, size_t cand_frame_stride,
                                                       int* __restrict__ cell_counts /* [frame][cell][2] */, int tile_bytes) {
    __shared__ int s_wave[4];
    __shared__ int s_base, s_n20;
    const int cell = blockIdx.y, frame = blockIdx.x;   // frame fastest: with a batch of 4 one frame's cells (and their halos) share two XCDs' L2s, with 8 one (describe_kernel)
    int lvl = 0;
    while (lvl + 1 < plan.nlevels && cell >= plan.lv[lvl + 1].cell_begin) ++lvl;
    const LevelDesc& L = plan.lv[lvl];
    const CellDesc C = cells[cell];
    const uint8_t* sc = score + (size_t)frame * frame_stride + L.img_off;
    uint32_t* out = cand + (size_t)frame * cand_frame_stride + L.cand_off + (size_t)(cell - L.cell_begin) * L.cellCap;
    const int iw = C.x1 - C.x0, ih = C.y1 - C.y0;
    const int npix = (iw > 0 && ih > 0 && !C.skipped) ? iw * ih : 0;
    const int minTh = plan.minTh, iniTh = plan.iniTh;
    if (threadIdx.x == 0) { s_base = 0; s_n20 = 0; }
    // stage the cell's strength values in LDS with one fully overlapped pass (all loads in flight together); cells larger
    // than the staging buffer (huge cells of tiny feature budgets) read HBM/L2 directly
    // Staging: the cell's strength values (s_tile) and one region shared by the image patch (FUSED) and the candidate lists, which never
    // live at the same time.  FUSED: dynamic LDS sized by the plan's largest cell (a 30 x 30 cell needs 5 KB, not the 36 KB of the
    // worst case: eight workgroups per CU instead of three — the kernel was occupancy bound from four frames on); the map form keeps
    // static worst-case arrays.
    extern __shared__ __attribute__((aligned(16))) uint8_t s_nms_dyn[];
    constexpr int kListInts = kNmsTileBytes / 2 + 4 * 64;
    __shared__ uint8_t s_tile_st[FUSED ? 16 : kNmsTileBytes];
    __shared__ __attribute__((aligned(16))) uint32_t s_un_st[FUSED ? 4 : kListInts];
    uint8_t* const s_tile = FUSED ? s_nms_dyn : s_tile_st;
    uint32_t* const s_un = FUSED ? reinterpret_cast<uint32_t*>(s_nms_dyn + tile_bytes) : s_un_st;
    const bool staged = FUSED ? npix > 0 : (npix > 0 && npix <= kNmsTileBytes);   // (FUSED plans only hold cells that fit)
    if constexpr (FUSED) {
        if (npix > 0) {
            uint8_t* s_img = reinterpret_cast<uint8_t*>(s_un);
            // the patch (cell + 3-pixel frame; always inside the level: a cell's sub-image is) as dwords from a 4-aligned column: ~2 loads
            // per thread instead of ~6 single bytes with their clamps.  LDS row stride pw, pixel (xx, yy) of the cell at column xx + 3 + shift.
            const int ph = ih + 6, xa = (C.x0 - 3) & ~3, shift = (C.x0 - 3) - xa;
            const int pw = (iw + 6 + shift + 3) & ~3, ndw = pw >> 2;
            const unsigned m_ndw = div_magic(ndw), m_iw = div_magic(iw);
            const uint8_t* pbase = sc + (size_t)(C.y0 - 3) * L.pitch + xa;
            for (int i = threadIdx.x; i < ndw * ph; i += 256) {
                const int py = div_by(i, ndw, m_ndw), c4 = i - py * ndw;
                reinterpret_cast<uint32_t*>(s_img)[i] = *reinterpret_cast<const u32_unaligned*>(pbase + (size_t)py * L.pitch + 4 * c4);
            }
            // the strength tile carries a one-pixel frame of zeros (stride iw + 2): the 3x3 maximum below reads its eight neighbours
            // without a bounds test; everything starts as 0 and only the pixels that are scored are written
            for (int i = threadIdx.x; i < ((iw + 2) * (ih + 2) + 3) / 4; i += 256) reinterpret_cast<uint32_t*>(s_tile)[i] = 0u;
            __syncthreads();
            // Only strengths >= minTh matter below (a candidate needs sv >= minTh, and a neighbour below minTh loses against it whatever
            // its exact value), so a pixel that cannot reach minTh is stored as 0 without being scored.  strength >= minTh needs an arc of
            // 9 contiguous circle pixels all darker than v - minTh (d > minTh) or all brighter (d < -minTh); an arc of 9 holds one pixel of
            // every opposite pair, so "one of (k, k + 8) is darker, for k = 0, 2, 4, 6" — or the same for brighter — is necessary: eight
            // of the sixteen reads and ~25 instructions instead of ~210.  The pixels that pass (a few per cent of an image) are queued per
            // wave and scored 64 at a time, so the full score runs on full waves.
            __shared__ uint16_t s_queue[4][128];
            const int lane_ = threadIdx.x & 63, wv_ = threadIdx.x >> 6;
            uint16_t* const queue = s_queue[wv_];
            auto score_at = [&](int i) {
                const int yy = div_by(i, iw, m_iw), xx = i - yy * iw;
                const uint8_t* c = s_img + (yy + 3) * pw + xx + 3 + shift;
                const int v = c[0];
                int d[25];
                d[0] = v - c[3 * pw + 0];   d[1] = v - c[3 * pw + 1];   d[2] = v - c[2 * pw + 2];
                d[3] = v - c[1 * pw + 3];   d[4] = v - c[3];            d[5] = v - c[-1 * pw + 3];
                d[6] = v - c[-2 * pw + 2];  d[7] = v - c[-3 * pw + 1];  d[8] = v - c[-3 * pw];
                d[9] = v - c[-3 * pw - 1];  d[10] = v - c[-2 * pw - 2]; d[11] = v - c[-1 * pw - 3];
                d[12] = v - c[-3];          d[13] = v - c[1 * pw - 3];  d[14] = v - c[2 * pw - 2];
                d[15] = v - c[3 * pw - 1];
#pragma unroll
                for (int k = 16; k < 25; k++) d[k] = d[k - 16];
                s_tile[(yy + 1) * (iw + 2) + xx + 1] = (uint8_t)max(fast_strength(d), 0);
            };
            int qn = 0;   // wave-uniform
            for (int i0 = wv_ * 64; i0 < npix; i0 += 256) {
                const int i = i0 + lane_;
                bool pass = false;
                if (i < npix) {
                    const int yy = div_by(i, iw, m_iw), xx = i - yy * iw;
                    const int gx = C.x0 + xx, gy = C.y0 + yy;
                    if (gx >= 3 && gy >= 3 && gx < L.w - 3 && gy < L.h - 3) {   // (fast_score_kernel's rule: the level's 3-pixel rim scores 0)
                        const uint8_t* c = s_img + (yy + 3) * pw + xx + 3 + shift;
                        const int v = c[0];
                        const int d0 = v - c[3 * pw], d8 = v - c[-3 * pw], d4 = v - c[3], d12 = v - c[-3];
                        const int d2 = v - c[2 * pw + 2], d10 = v - c[-2 * pw - 2], d6 = v - c[-2 * pw + 2], d14 = v - c[2 * pw - 2];
                        const int dark = min(min(max(d0, d8), max(d4, d12)), min(max(d2, d10), max(d6, d14)));
                        const int bright = max(max(min(d0, d8), min(d4, d12)), max(min(d2, d10), min(d6, d14)));
                        pass = dark > minTh || bright < -minTh;
                    }
                }
                const unsigned long long m = __ballot(pass);
                if (pass) queue[qn + __popcll(m & ((1ull << lane_) - 1ull))] = (uint16_t)i;
                qn += __popcll(m);
                if (qn >= 64) {
                    qn -= 64;
                    score_at(queue[qn + lane_]);
                }
            }
            if (lane_ < qn) score_at(queue[lane_]);
        }
    } else {
        if (staged)
            for (int i = threadIdx.x; i < npix; i += 256) { const int yy = i / iw, xx = i - yy * iw; s_tile[i] = sc[(size_t)(C.y0 + yy) * L.pitch + C.x0 + xx]; }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (staged) {
        // Fast path: each wave owns one contiguous quarter of the cell's raster, compacts its candidates into its own LDS
        // list with no workgroup barrier inside the loop, then the four lists are concatenated in wave (= raster) order.
        uint32_t* const s_list = s_un;
        __shared__ int s_cnt[4], s_c20[4];
        const unsigned m_iw2 = div_magic(iw);
        const int chunk = ((npix + 3) / 4 + 63) & ~63;            // pixels per wave, multiple of 64
        uint32_t* mylist = s_list + wv * (chunk / 2 + 64);        // NMS packing bound: <= every other pixel of a range
        const int pbeg = wv * chunk, pend = min(pbeg + chunk, npix);
        int k = 0, k20 = 0;
        for (int p0 = pbeg; p0 < pend; p0 += 64) {
            const int pp = p0 + lane;
            const bool live = pp < pend;
            const int pc = live ? pp : pbeg;
            const int yy = div_by(pc, iw, m_iw2), xx = pc - yy * iw;
            int sv, nmax = 0;
            if constexpr (FUSED) {   // zero-framed tile
                const int tw = iw + 2;
                const uint8_t* c = s_tile + (yy + 1) * tw + xx + 1;
                sv = c[0];
                nmax = max(max(max((int)c[-tw - 1], (int)c[-tw]), max((int)c[-tw + 1], (int)c[-1])), max(max((int)c[1], (int)c[tw - 1]), max((int)c[tw], (int)c[tw + 1])));
            } else {
                sv = s_tile[pc];
#pragma unroll
                for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                    for (int dx = -1; dx <= 1; dx++) {
                        if (dx == 0 && dy == 0) continue;
                        const bool in = xx + dx >= 0 && xx + dx < iw && yy + dy >= 0 && yy + dy < ih;
                        const int nv = s_tile[in ? pc + dy * iw + dx : pc];
                        nmax = max(nmax, in ? nv : 0);
                    }
            }
            const bool keep = live && sv >= minTh && sv > nmax;
            const unsigned long long m = __ballot(keep);
            if (keep) mylist[k + __popcll(m & ((1ull << lane) - 1ull))] = ((uint32_t)sv << 24) | ((uint32_t)(C.y0 + yy) << 12) | (uint32_t)(C.x0 + xx);
            k += __popcll(m);
            k20 += __popcll(__ballot(keep && sv >= iniTh));
        }
        if (lane == 0) { s_cnt[wv] = k; s_c20[wv] = k20; }
        __syncthreads();
        int off = 0;
        for (int i = 0; i < wv; i++) off += s_cnt[i];
        for (int i = lane; i < k; i += 64) if (off + i < L.cellCap) out[off + i] = mylist[i];
        if (threadIdx.x == 0) {
            int* ccw = cell_counts + ((size_t)frame * plan.total_cells + cell) * 2;
            ccw[0] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
            ccw[1] = s_c20[0] + s_c20[1] + s_c20[2] + s_c20[3];
        }
        return;
    }
    // Large-cell path (strength values read from HBM/L2).: raster order = (thread, j) lexicographic, ranked with four ballots
    for (int p0 = 0; p0 < npix; p0 += 1024) {
        const int pbase = p0 + threadIdx.x * 4;
        int yy = 0, xx = 0;
        if (pbase < npix) { yy = pbase / iw; xx = pbase - yy * iw; }
        bool keep[4];
        uint32_t word[4];
        int n20_local = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            keep[j] = false;
            word[j] = 0;
            if (pbase + j < npix) {
                while (xx >= iw) { xx -= iw; ++yy; }
                const int x = C.x0 + xx, y = C.y0 + yy;
                const int sv = staged ? (int)s_tile[yy * iw + xx] : (int)sc[(size_t)y * L.pitch + x];
                // all eight neighbours are fetched unconditionally (clamped address, masked value) so that the reads overlap;
                // a short-circuit chain would serialise eight dependent LDS round trips per candidate
                int nmax = 0;
#pragma unroll
                for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                    for (int dx = -1; dx <= 1; dx++) {
                        if (dx == 0 && dy == 0) continue;
                        const int nx = x + dx, ny = y + dy;
                        const bool in = nx >= C.x0 && nx < C.x1 && ny >= C.y0 && ny < C.y1;
                        const int cy = in ? yy + dy : yy, cx = in ? xx + dx : xx;
                        const int nv = staged ? (int)s_tile[cy * iw + cx] : (int)sc[(size_t)(C.y0 + cy) * L.pitch + C.x0 + cx];
                        nmax = max(nmax, in ? nv : 0);
                    }
                const bool k = sv >= minTh && sv > nmax;
                keep[j] = k;
                word[j] = ((uint32_t)sv << 24) | ((uint32_t)y << 12) | (uint32_t)x;
                if (k && sv >= iniTh) n20_local++;
            }
            ++xx;
        }
        const unsigned long long lt = (1ull << lane) - 1ull;
        int before = 0, wave_total = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned long long m = __ballot(keep[j]);
            before += __popcll(m & lt);
            wave_total += __popcll(m);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) n20_local += __shfl_xor(n20_local, o);
        if (lane == 0) { s_wave[wv] = wave_total; if (n20_local) atomicAdd(&s_n20, n20_local); }
        __syncthreads();
        int pos = s_base + before;
        for (int i = 0; i < wv; i++) pos += s_wave[i];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (keep[j]) { if (pos < L.cellCap) out[pos] = word[j]; ++pos; }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int* cc = cell_counts + ((size_t)frame * plan.total_cells + cell) * 2;
        cc[0] = s_base;
        cc[1] = s_n20;
    }
}

struct SelectArgs {
    const LevelDesc* L;
    const int* cc;
    const uint32_t* cbase;
    const int* s_nkeys; const int* s_ret; const int* s_off; const int* s_out;
    int nCells, iniTh;
    int c_lo, c_hi;     // this workgroup's cells
    int scratch_ints;   // LDS path: ints available behind the cells' lists for the partition scratch (0 = use the sequential routine)
};

// Phase 1, this workgroup's cells [c_lo, c_hi): per-cell retainBest + truncate (:1053-1055); the survivors go to their place in the level's
// concatenation (:1058-1065) in HBM.  WorkPtr is either an LDS or a global pointer (static address space); `work` addresses the
// filtered list of cell c at s_off[c] - s_off[c_lo].  Cells are spread over the waves; each selection is wave-cooperative
// (introselect.hpp) when the LDS scratch allows it and falls back to the sequential routine otherwise (same data movement either way).
template <typename WorkPtr>
__device__ __forceinline__ void select_cells(WorkPtr work, int* scratch, uint32_t* __restrict__ cat_g, const SelectArgs& A) {
    const LevelDesc& L = *A.L;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int per_wave = A.scratch_ints / kSelWaves;     // each wave: Lpos | Rpos halves
    int* myL = scratch + wv * per_wave;
    int* myR = myL + per_wave / 2;
    const int off0 = A.s_off[A.c_lo];
    for (int c = A.c_lo + wv; c < A.c_hi; c += kSelWaves) {
        const int nk = A.s_nkeys[c], ret = A.s_ret[c];
        if (nk == 0) continue;
        const uint32_t* src = A.cbase + (size_t)c * L.cellCap;
        WorkPtr w = work + (A.s_off[c] - off0);
        const int n7 = A.cc[2 * c], n20 = A.cc[2 * c + 1];
        // threshold filter of the cell's raster-ordered candidates (:980-987), order preserving
        int k = 0;
        for (int i0 = 0; i0 < n7; i0 += 64) {
            const int i = i0 + lane;
            const uint32_t e = i < n7 ? src[i] : 0u;
            const bool keep = i < n7 && (n20 > 3 ? (int)(e >> 24) >= A.iniTh : true);
            const unsigned long long m = __ballot(keep);
            if (keep) w[k + __popcll(m & ((1ull << lane) - 1ull))] = e;
            k += __popcll(m);
        }
        uh_sel::wave_mem_sync();
        // KeyPointsFilter::retainBest(keysCell, ret) then resize(ret): only the nth_element data movement matters
        if (ret > 0 && nk > ret) {
            if (per_wave / 2 >= nk) uh_sel::wave_nth_element_desc(w, nk, ret - 1, myL, myR, lane);
            else { if (lane == 0) uh_sel::nth_element_desc(w, nk, ret - 1); uh_sel::wave_mem_sync(); }
        }
        uint32_t* o = cat_g + A.s_out[c];   // write-through stores: the level's last workgroup may sit on another XCD (non-coherent L2)
        for (int i = lane; i < ret; i += 64) __hip_atomic_store(o + i, (uint32_t)w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Phase 2, the workgroup that finishes a level last: level-wide retainBest + truncate (:1069-1073) over the concatenation, then
// computeDescriptors' border filter (:1124-1130), order preserving (the reference applies it before describing).
template <typename CatPtr>
__device__ __forceinline__ void select_level(CatPtr cat, int* scratch, int scratch_ints, int total, const LevelDesc& L, uint32_t* __restrict__ lsel,
                                             int* __restrict__ level_count) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wv != 0) return;
    if (total > L.nDesired) {
        if (L.nDesired > 0) {
            if (scratch_ints / 2 >= total) uh_sel::wave_nth_element_desc(cat, total, L.nDesired - 1, scratch, scratch + scratch_ints / 2, lane);
            else { if (lane == 0) uh_sel::nth_element_desc(cat, total, L.nDesired - 1); uh_sel::wave_mem_sync(); }
        }
        total = L.nDesired;
    }
    const int maxX = L.w - EDGE, maxY = L.h - EDGE;
    int k = 0;
    for (int i0 = 0; i0 < total; i0 += 64) {
        const int i = i0 + lane;
        const uint32_t e = i < total ? cat[i] : 0u;
        const int x = e & 0xFFF, y = (e >> 12) & 0xFFF;
        const bool keep = i < total && !(x < EDGE || y < EDGE || x > maxX || y > maxY);
        const unsigned long long m = __ballot(keep);
        if (keep) lsel[k + __popcll(m & ((1ull << lane) - 1ull))] = e;
        k += __popcll(m);
    }
    if (lane == 0) *level_count = k;
}

// ------------------------------------------------------------------------------------------------ selection
// Several workgroups per (frame, level) — one per 64 cells: every one of them repeats the level's quota redistribution (:994-1039, a few
// parallel sweeps), then selects inside ITS cells (per-cell retainBest + truncate, :1053-1055) and writes the survivors to their place
// in the level's concatenation (cell-row-major order, :1058-1065).  The workgroup that finishes a level last (a ticket counter, nobody
// waits) runs the level-wide retainBest + truncate (:1069-1073) and the border filter.  One workgroup per level spent 40 of its 49 us
// walking level 0's ~490 cells with 16 waves.
__global__ __launch_bounds__(kSelThreads) void select_kernel(const Plan plan, const CellDesc* __restrict__ cells,
                                                     const uint32_t* __restrict__ cand, size_t cand_frame_stride,
                                                     const int* __restrict__ cell_counts, uint32_t* __restrict__ work_g,
                                                     size_t work_frame_stride, uint32_t* __restrict__ sel,
                                                     size_t sel_frame_stride, int* __restrict__ level_counts, int lds_entries,
                                                     int* __restrict__ tickets) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    __shared__ int s_nkeys[kMaxCellsPerLevel];
    __shared__ int s_ret[kMaxCellsPerLevel];
    __shared__ int s_off[kMaxCellsPerLevel + 1];
    __shared__ int s_out[kMaxCellsPerLevel + 1];
    __shared__ int s_total;
    const int frame = blockIdx.y;
    int lvl = 0;
    while (lvl + 1 < plan.nlevels && (int)blockIdx.x >= plan.sel_wg_begin[lvl + 1]) ++lvl;
    const int part = blockIdx.x - plan.sel_wg_begin[lvl], nparts = plan.sel_wg_begin[lvl + 1] - plan.sel_wg_begin[lvl];
    const LevelDesc& L = plan.lv[lvl];
    const int nCells = L.nCells;
    const int* cc = cell_counts + ((size_t)frame * plan.total_cells + L.cell_begin) * 2;
    const uint32_t* cbase = cand + (size_t)frame * cand_frame_stride + L.cand_off;
    uint32_t* lsel = sel + (size_t)frame * sel_frame_stride + L.sel_off;
    const int iniTh = plan.iniTh;

    __shared__ unsigned char s_skipped[kMaxCellsPerLevel];
    for (int c = threadIdx.x; c < nCells; c += kSelThreads) {
        const int n7 = cc[2 * c], n20 = cc[2 * c + 1];
        const int sk = cells[L.cell_begin + c].skipped;
        s_skipped[c] = (unsigned char)sk;
        s_nkeys[c] = sk ? 0 : (n20 > 3 ? n20 : n7);   // :980-987
    }
    __syncthreads();
    // Quota redistribution (:989-1039) and the two exclusive scans, one thread per cell (round 1 ran them on thread 0: ~80 cells x 3-4
    // sweeps of dependent LDS reads = half of this kernel's 60 us).  A sweep's decision for a cell depends only on the sweep's quota and
    // the cell's own count, and the sweep's results are integer sums — the order of the cells does not matter.
    __shared__ int s_ws[2][kSelWaves];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto block_sum2 = [&](int a, int b, int& ra, int& rb) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        __syncthreads();   // previous readers of s_ws are done; the cell flags written above are visible afterwards
        if (lane == 0) { s_ws[0][wave] = a; s_ws[1][wave] = b; }
        __syncthreads();
        ra = 0; rb = 0;
#pragma unroll
        for (int w = 0; w < kSelWaves; w++) { ra += s_ws[0][w]; rb += s_ws[1][w]; }
    };
    const int nfeaturesCell = L.nfeaturesCell;
    int nNoMore = 0, nToDistribute = 0;
    {
        int dist = 0, nom = 0;   // first pass: skipped cells are not visited (their bNoMore stays false); s_off doubles as bNoMore
        for (int c = threadIdx.x; c < nCells; c += kSelThreads) {
            if (s_skipped[c]) { s_ret[c] = 0; s_off[c] = 0; continue; }
            const int nKeys = s_nkeys[c];
            if (nKeys > nfeaturesCell) { s_ret[c] = nfeaturesCell; s_off[c] = 0; }
            else { s_ret[c] = nKeys; dist += nfeaturesCell - nKeys; s_off[c] = 1; nom++; }
        }
        block_sum2(dist, nom, nToDistribute, nNoMore);
    }
    while (nToDistribute > 0 && nNoMore < nCells) {   // (uniform: every thread holds the same sums)
        const int nNew = (int)((float)nfeaturesCell + ceilf((float)nToDistribute / (float)(nCells - nNoMore)));
        int dist = 0, nom = 0;
        for (int c = threadIdx.x; c < nCells; c += kSelThreads) {
            if (s_off[c]) continue;
            const int nTotal = s_nkeys[c];
            if (nTotal > nNew) { s_ret[c] = nNew; }
            else { s_ret[c] = nTotal; dist += nNew - nTotal; s_off[c] = 1; nom++; }
        }
        int sd, sn;
        block_sum2(dist, nom, sd, sn);
        nToDistribute = sd;
        nNoMore += sn;
    }
    __syncthreads();
    {   // exclusive scans of the candidate counts (-> s_off) and of the quotas (-> s_out): two cells per thread
        const int c0 = 2 * threadIdx.x, c1 = c0 + 1;
        const int a0 = c0 < nCells ? s_nkeys[c0] : 0, a1 = c1 < nCells ? s_nkeys[c1] : 0;
        const int b0 = c0 < nCells ? s_ret[c0] : 0, b1 = c1 < nCells ? s_ret[c1] : 0;
        int ia = a0 + a1, ib = b0 + b1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int ua = __shfl_up(ia, o), ub = __shfl_up(ib, o);
            if (lane >= o) { ia += ua; ib += ub; }
        }
        __syncthreads();   // everybody has read s_ret / s_nkeys and the bNoMore flags in s_off are dead
        if (lane == 63) { s_ws[0][wave] = ia; s_ws[1][wave] = ib; }
        __syncthreads();
        int pa = 0, pb = 0, ta = 0, tb = 0;
#pragma unroll
        for (int w = 0; w < kSelWaves; w++) {
            if (w < wave) { pa += s_ws[0][w]; pb += s_ws[1][w]; }
            ta += s_ws[0][w]; tb += s_ws[1][w];
        }
        const int ea = pa + ia - (a0 + a1), eb = pb + ib - (b0 + b1);
        if (c0 < nCells) { s_off[c0] = ea; s_out[c0] = eb; }
        if (c1 < nCells) { s_off[c1] = ea + a0; s_out[c1] = eb + b0; }
        if (threadIdx.x == 0) { s_off[nCells] = ta; s_out[nCells] = tb; s_total = tb; }
    }
    __syncthreads();
    // ---- phase 1: this workgroup's cells.  Workspace: LDS when their filtered candidates fit, else the level's HBM scratch (the two
    // calls are separate instantiations so that the pointer's address space is static: a generic pointer would compile every access of
    // the selection into a flat_load).
    uint32_t* const work_lvl = work_g + (size_t)frame * work_frame_stride + 2 * (size_t)L.cand_off;   // [filtered lists | concatenation]
    uint32_t* const cat_g = work_lvl + s_off[nCells];
    const int chunk = (nCells + nparts - 1) / nparts;
    SelectArgs A{&L, cc, cbase, s_nkeys, s_ret, s_off, s_out, nCells, iniTh, min(part * chunk, nCells), min((part + 1) * chunk, nCells), 0};
    const int need = s_off[A.c_hi] - s_off[A.c_lo];
    if (need <= lds_entries) {
        A.scratch_ints = ((lds_entries - need) / (2 * kSelWaves)) * (2 * kSelWaves);   // whatever LDS is left becomes partition scratch
        select_cells(s_dyn, reinterpret_cast<int*>(s_dyn + need), cat_g, A);
    } else {
        select_cells(work_lvl + s_off[A.c_lo], static_cast<int*>(nullptr), cat_g, A);   // sequential selection in HBM
    }
    // ---- ticket: the last workgroup of the level goes on
    __shared__ int s_ticket;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's write-through stores have landed before the ticket is drawn
    __syncthreads();                                   // (a device-wide fence here costs a whole L2 write-back: 6 us, measured)
    int* const ticket = tickets + (size_t)frame * kMaxLevels + lvl;
    if (threadIdx.x == 0) s_ticket = nparts > 1 ? atomicAdd(ticket, 1) : 0;
    __syncthreads();
    if (s_ticket != nparts - 1) return;
    if (threadIdx.x == 0 && nparts > 1) *ticket = 0;   // ready for the next launch
    // ---- phase 2: the whole level
    const int total = s_total;
    int* const lc = level_counts + (size_t)frame * kMaxLevels + lvl;
    if (total <= lds_entries) {
        for (int i = threadIdx.x; i < total; i += kSelThreads) s_dyn[i] = __hip_atomic_load(cat_g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (L2-bypassing)
        __syncthreads();
        select_level(s_dyn, reinterpret_cast<int*>(s_dyn + total), ((lds_entries - total) / 2) * 2, total, L, lsel, lc);
    } else {
        // levels too large for LDS are selected in HBM with plain loads: the other workgroups' write-through stores are in memory, but
        // this CU's L1 / this XCD's L2 may still hold lines of cat_g from before they were written — invalidate them first
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        select_level(cat_g, static_cast<int*>(nullptr), 0, total, L, lsel, lc);
    }
}

// ------------------------------------------------------------------------------------------------ orientation + rBRIEF
struct KeyPointOut { float x, y, size, angle, response; int octave, class_id; };   // == cv::KeyPoint (28 bytes)

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {   // cv::fastAtan2 scalar path, un-fused
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI), p3 = -0.3258083974640975f * (float)(180 / M_PI);
    const float p5 = 0.1555786518463281f * (float)(180 / M_PI), p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------------------------------------ optional radius-3 NMS
// ORBextractor::processLevel with the debug switch "orb_nonmaxima" (ORBextractor.cpp:1146-1148,1176-1205): after the per-level
// keypoint selection and before the descriptors, visit the level's keypoints IN ORDER; a keypoint that is still a possible
// maximum looks at every keypoint within radius 3 of it (squared distance < 9, itself included; suppressed ones count too)
// and marks all of them whose response is below the largest response found there.  The visiting order matters (a keypoint
// suppressed before its turn never suppresses its own neighbourhood), so this is a sequential walk: one wave per (level,
// frame), the 64 lanes scan the level's list for the disc of the current keypoint.  picoflann's radius search returns
// exactly the brute-force disc (tests/test_projmatch_oracle.py), and max / mark-all do not depend on the order inside it.
__global__ __launch_bounds__(64) void nonmax_kernel(const Plan plan, uint32_t* __restrict__ sel, size_t sel_frame_stride,
                                                    int* __restrict__ level_counts) {
    extern __shared__ uint32_t s_list[];   // n entries, then n class bytes
    const int lvl = blockIdx.x, frame = blockIdx.y, lane = threadIdx.x;
    int* cnt = level_counts + (size_t)frame * kMaxLevels + lvl;
    const int n = *cnt;
    if (n <= 0) return;
    uint32_t* list = sel + (size_t)frame * sel_frame_stride + plan.lv[lvl].sel_off;
    unsigned char* s_cls = reinterpret_cast<unsigned char*>(s_list + n);
    for (int i = lane; i < n; i += 64) { s_list[i] = list[i]; s_cls[i] = 1; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int i = 0; i < n; i++) {
        if (!s_cls[i]) continue;                       // wave-uniform
        const uint32_t ei = s_list[i];
        const int xi = ei & 0xFFF, yi = (ei >> 12) & 0xFFF;
        int mx = 0;                                    // int maxResponse = 0
        for (int j = lane; j < n; j += 64) {
            const uint32_t e = s_list[j];
            const int dx = xi - (int)(e & 0xFFF), dy = yi - (int)((e >> 12) & 0xFFF);
            if (dx * dx + dy * dy < 9) mx = max(mx, (int)(e >> 24));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
        for (int j = lane; j < n; j += 64) {
            const uint32_t e = s_list[j];
            const int dx = xi - (int)(e & 0xFFF), dy = yi - (int)((e >> 12) & 0xFFF);
            if (dx * dx + dy * dy < 9 && (int)(e >> 24) < mx) s_cls[j] = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // std::remove_if: stable compaction
    int kept = 0;
    for (int j0 = 0; j0 < n; j0 += 64) {
        const int j = j0 + lane;
        const bool keep = j < n && s_cls[j] != 0;
        const unsigned long long m = __ballot(keep);
        if (keep) list[kept + __popcll(m & ((1ull << lane) - 1ull))] = s_list[j];
        kept += __popcll(m);
    }
    if (lane == 0) *cnt = kept;
}

__device__ __forceinline__ void describe_slot(const Plan& plan, const uint8_t* __restrict__ pyr, size_t frame_stride, const uint32_t* __restrict__ sel,
                                              size_t sel_frame_stride, const int* __restrict__ lc, KeyPointOut* s_kp,
                                              unsigned long long* s_desc, int class_id, int frame, int lane, int slot);

// Camera model of the frames (ImageParams: CameraMatrix CV_32F fx fy cx cy, Distorsion k1 k2 p1 p2 [k3 [k4 k5 k6]]) for the undistorted
// keypoints: undistortPoints(points, ImageParams) of src/basictypes/misc.cpp:269-293 = cv::undistortPoints(points, out, K, D) — normalise
// with 1/fx, 1/fy in double, FIVE fixed-point iterations of the distortion model (the overload's TermCriteria(MAX_ITER, 5, 0.01): a count,
// no early exit), a float result — followed by x * fx + cx in FLOAT arithmetic (misc.cpp:283-291; this unit is compiled without
// contraction).  Called for every frame's keypoints at src/utils/frameextractor.cpp:3985; parity UNPINNED like the rest of the extractor
// (OpenCV absent): KATs + oracle/orb_oracle.cpp::oracle_undistort_points.
struct CamModel { double fx, fy, cx, cy, ifx, ify, k[8]; float ffx, ffy, fcx, fcy; int on; };
__host__ __device__ inline void undistort_point(const CamModel& c, float u, float v, float& ox, float& oy) {
    double x = ((double)u - c.cx) * c.ifx, y = ((double)v - c.cy) * c.ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((c.k[7] * r2 + c.k[6]) * r2 + c.k[5]) * r2) / (1 + ((c.k[4] * r2 + c.k[1]) * r2 + c.k[0]) * r2);
        if (icdist < 0) { x = x0; y = y0; break; }   // (OpenCV: the point is left at its normalised position)
        const double dX = 2 * c.k[2] * x * y + c.k[3] * (r2 + 2 * x * x);
        const double dY = c.k[2] * (r2 + 2 * y * y) + 2 * c.k[3] * x * y;
        x = (x0 - dX) * icdist;
        y = (y0 - dY) * icdist;
    }
    ox = (float)x * c.ffx + c.fcx;
    oy = (float)y * c.ffy + c.fcy;
}

// One wave per output keypoint slot; 4 waves per block.
__global__ __launch_bounds__(256) void describe_kernel(const Plan plan, const uint8_t* __restrict__ pyr,
                                                       size_t frame_stride, const uint32_t* __restrict__ sel,
                                                       size_t sel_frame_stride, const int* __restrict__ level_counts,
                                                       KeyPointOut* __restrict__ kps, uint8_t* __restrict__ desc,
                                                       int cap_per_frame, int* __restrict__ frame_counts, int class_id,
                                                       const CamModel cam, float* __restrict__ und, int batch,
                                                       uint8_t* __restrict__ desc_dev, float4* __restrict__ kd_in) {
    // 1-D grid, frame fastest: consecutive workgroups go to consecutive XCDs, so with a batch of 8 one frame's pyramid is pulled into ONE
    // XCD's L2 (batch 4, the bench step: two) instead of all eight (2-D grid, round 1-4: FETCH_SIZE x2 22.7 MB per 4-frame launch
    // against 7.6 MB of blurred pyramid; profiles/r05_fetch_size_calibration.json for the counter's meaning on byte gathers)
    const int frame = blockIdx.x % batch, bx = blockIdx.x / batch;
    const int lane = threadIdx.x & 63;
    const int slot = __builtin_amdgcn_readfirstlane(bx * 4 + (threadIdx.x >> 6));
    const int* lc = level_counts + (size_t)frame * kMaxLevels;
    int total = 0;
    for (int l = 0; l < plan.nlevels; l++) total += lc[l];
    if (slot == 0 && lane == 0) frame_counts[frame] = total;
    // the workgroup's four keypoints are staged in LDS and leave as 16-byte stores: 8 + 7 store instructions per workgroup instead of
    // 16 eight-byte and 8 struct pieces — what the write path to pinned host memory (one PCIe transaction per request) is bound by
    __shared__ __attribute__((aligned(16))) unsigned long long s_desc[4][4];
    __shared__ __attribute__((aligned(16))) KeyPointOut s_kp[4];
    const int wv = threadIdx.x >> 6;
    const int limit = total < cap_per_frame ? total : cap_per_frame;
    if (slot < limit) describe_slot(plan, pyr, frame_stride, sel, sel_frame_stride, lc, &s_kp[wv], s_desc[wv], class_id, frame, lane, slot);
    __syncthreads();
    {
        const int slot0 = bx * 4, nvalid = min(max(limit - slot0, 0), 4);
        const size_t o = (size_t)frame * cap_per_frame + slot0;
        const int t = threadIdx.x;
        if (t < 2 * nvalid) {
            const uint4 w = reinterpret_cast<const uint4*>(&s_desc[0][0])[t];
            reinterpret_cast<uint4*>(desc + o * 32)[t] = w;
            if (desc_dev) reinterpret_cast<uint4*>(desc_dev + o * 32)[t] = w;   // the frame stays on the device too (uh_dev_frame: the projection matcher's copy)
        } else if (und && t >= 128 && t < 128 + nvalid) {   // (a third wave) the undistorted position of each of the four keypoints, Frame::und_kpts
            float ux, uy;
            undistort_point(cam, s_kp[t - 128].x, s_kp[t - 128].y, ux, uy);
            reinterpret_cast<float2*>(und)[o + (t - 128)] = make_float2(ux, uy);
            if (kd_in) kd_in[o + (t - 128)] = make_float4(ux, uy, __int_as_float(s_kp[t - 128].octave), 0.f);   // input of the kd-tree build launch (kdbuild.hip)
        }
        if (t >= 64 && t < 64 + 7 * nvalid) {   // (a second wave: the two groups of stores issue side by side)
            const int i = t - 64;
            uint32_t* dst = reinterpret_cast<uint32_t*>(kps + o);
            if (nvalid == 4 && ((o * sizeof(KeyPointOut)) & 15) == 0 && (reinterpret_cast<uintptr_t>(kps) & 15) == 0) {
                if (i < 7) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(&s_kp[0])[i];
            } else dst[i] = reinterpret_cast<const uint32_t*>(&s_kp[0])[i];
        }
    }
}

__device__ __forceinline__ void describe_slot(const Plan& plan, const uint8_t* __restrict__ pyr, size_t frame_stride, const uint32_t* __restrict__ sel,
                                              size_t sel_frame_stride, const int* __restrict__ lc, KeyPointOut* s_kp,
                                              unsigned long long* s_desc, int class_id, int frame, int lane, int slot) {
    int lvl = 0, base = 0;
    while (slot >= base + lc[lvl]) { base += lc[lvl]; ++lvl; }
    // the level is wave-uniform (slot is), but the compiler cannot see that: without the readfirstlane it copied the whole by-value Plan
    // to scratch to index it per lane — 120 bytes of scratch per lane, the "37x write amplification" of this kernel in the round-1 counters
    lvl = __builtin_amdgcn_readfirstlane(lvl);
    base = __builtin_amdgcn_readfirstlane(base);
    const LevelDesc& L = plan.lv[lvl];
    const uint32_t e = sel[(size_t)frame * sel_frame_stride + L.sel_off + (slot - base)];
    const int cx = e & 0xFFF, cy = (e >> 12) & 0xFFF, resp = e >> 24;
    const uint8_t* img = pyr + (size_t)frame * frame_stride + L.img_off;
    const uint8_t* center = img + (size_t)cy * L.pitch + cx;
    // IC_Angle: integer moments over the radius-15 disc, one row per lane
    int m10 = 0, m01 = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) {   // (integer sums: the order of the pixels does not matter)
        const int uv = d_disc.uv[k * 64 + lane];
        const int u = (int)(signed char)(uv & 255), v = uv >> 8;
        const int val = center[(ptrdiff_t)v * L.pitch + u];
        m10 += u * val;
        m01 += v * val;
    }
    m10 = wave_sum(m10);
    m01 = wave_sum(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    // rotated BRIEF: lane t evaluates tests 4t..4t+3
    const float factorPI = (float)(M_PI / 180.f);
    const float ang = angle * factorPI;
    // libm's cosf/sinf, bit for bit (glibc_sincosf.hpp): a correctly rounded cosine differs from them in the last bit often
    // enough to flip a descriptor bit every few million descriptors
    const float a = uh_sincosf::cosf_glibc(ang), b = uh_sincosf::sinf_glibc(ang);
    // lane t evaluates tests t, 64 + t, 128 + t, 192 + t: the ballot of round j IS descriptor word j (bit i of byte b = test 8b + i,
    // ORBextractor.cpp:129-150), so a keypoint's 32 bytes leave as four 8-byte stores from lanes 0..3 instead of 32 single-byte stores
    // (round 1: WRITE_SIZE 37x the algorithmic bytes of this kernel)
    unsigned long long word[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const signed char* p = d_pattern + (64 * j + lane) * 4;
        const int x0 = p[0], y0 = p[1], x1 = p[2], y1 = p[3];
        const int r0 = (int)rintf((float)x0 * b + (float)y0 * a), c0 = (int)rintf((float)x0 * a - (float)y0 * b);
        const int r1 = (int)rintf((float)x1 * b + (float)y1 * a), c1 = (int)rintf((float)x1 * a - (float)y1 * b);
        const int t0 = center[(ptrdiff_t)r0 * L.pitch + c0], t1 = center[(ptrdiff_t)r1 * L.pitch + c1];
        word[j] = __ballot(t0 < t1);
    }
    // into the workgroup's staging block: the four keypoints of a workgroup leave together as 16-byte stores (describe_kernel)
    if (lane < 4) {
        const unsigned long long w = lane == 0 ? word[0] : (lane == 1 ? word[1] : (lane == 2 ? word[2] : word[3]));
        s_desc[lane] = w;
    }
    if (lane == 0) {
        KeyPointOut k;
        k.x = (float)cx; k.y = (float)cy;
        if (lvl != 0) { k.x = (k.x + 0.5f) * L.scale; k.y = (k.y + 0.5f) * L.scale; }   // :1228-1229
        k.size = (float)L.scaledPatchSize;
        k.angle = angle;
        k.response = (float)resp;
        k.octave = lvl;
        k.class_id = class_id;   // -1; 1 when the radius-3 suppression ran (it uses class_id as its flag and leaves it, :1179)
        *s_kp = k;
    }
}

// uh_orb_extract_frame_dev_begin: the frame's undistorted keypoints (position + octave: what Frame::create_kdtree needs) leave for the host
// as soon as the selection is done — one workgroup, behind select_kernel and in front of describe_kernel — with a completion word of their
// own, so that the host builds the kd-tree while the descriptors are still being computed.  Same float expressions as describe_slot /
// describe_kernel's third wave (the final values are written again there, identically).
struct EarlyUnd { KeyPointOut* kpts; int* count; unsigned long long* word_ptr; unsigned long long word; int cap; unsigned* ticket; };
constexpr int kEarlyThreads = 256, kEarlyGroups = 8;   // (cv::undistortPoints' five double-precision iterations for 2000 points on ONE compute unit were 7 us)
__global__ __launch_bounds__(kEarlyThreads) void und_early_kernel(const Plan plan, const uint32_t* __restrict__ sel, const int* __restrict__ lc, const CamModel cam, const EarlyUnd ea) {
    __shared__ int s_base[kMaxLevels + 1], s_off[kMaxLevels];
    __shared__ float s_scale[kMaxLevels];
    __shared__ unsigned s_last;
    // 256 records at a time through LDS: they leave as 16-byte stores of consecutive lanes (28-byte records written field by field are one small
    // host-link write each)
    __shared__ __attribute__((aligned(16))) KeyPointOut s_rec[kEarlyThreads];
    const int tid = threadIdx.x;
    if (tid == 0) {
        int b = 0;
        for (int l = 0; l < plan.nlevels; l++) { s_base[l] = b; b += lc[l]; }
        for (int l = plan.nlevels; l <= kMaxLevels; l++) s_base[l] = b;
    }
    if (tid < plan.nlevels) { s_off[tid] = plan.lv[tid].sel_off; s_scale[tid] = plan.lv[tid].scale; }
    __syncthreads();
    const int total = s_base[plan.nlevels], limit = total < ea.cap ? total : ea.cap;
    for (int s0 = (int)blockIdx.x * kEarlyThreads; s0 < limit; s0 += (int)gridDim.x * kEarlyThreads) {
        const int slot = s0 + tid;
        if (slot < limit) {
            int lvl = 0;
            while (slot >= s_base[lvl + 1]) ++lvl;
            const uint32_t e = sel[s_off[lvl] + (slot - s_base[lvl])];
            const int cx = e & 0xFFF, cy = (e >> 12) & 0xFFF;
            float kx = (float)cx, ky = (float)cy;
            if (lvl != 0) { kx = (kx + 0.5f) * s_scale[lvl]; ky = (ky + 0.5f) * s_scale[lvl]; }
            float ux, uy;
            undistort_point(cam, kx, ky, ux, uy);
            KeyPointOut k;
            k.x = ux; k.y = uy; k.size = 0.f; k.angle = 0.f; k.response = 0.f; k.octave = lvl; k.class_id = -1;
            s_rec[tid] = k;
        }
        __syncthreads();
        const int cnt = min(kEarlyThreads, limit - s0);
        const int n16 = (cnt * (int)sizeof(KeyPointOut) + 15) / 16;   // (s0 * 28 is a multiple of 16: 256 * 28 = 16 * 448; the tail's padding bytes stay inside the block)
        uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<char*>(ea.kpts) + (size_t)s0 * sizeof(KeyPointOut));
        const uint4* src = reinterpret_cast<const uint4*>(s_rec);
        for (int i = tid; i < n16; i += kEarlyThreads) dst[i] = src[i];
        __syncthreads();
    }
    // every workgroup releases its stores into pinned memory, then takes a ticket; the last one posts the count and the word
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(ea.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (s_last && tid == 0) {
        __hip_atomic_store(ea.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *ea.count = total;
        __hip_atomic_store(ea.word_ptr, ea.word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------------------ host side
inline int cvRoundf(float v) { return (int)lrintf(v); }
inline int cvFloord(double v) { int i = (int)v; return i - (i > v); }

void interpolate_cubic(float x, float* c) {
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

void cubic_taps(int ssize, int dsize, std::vector<int>& ofs, std::vector<short>& coef) {
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1.0 / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cvFloord(f);
        f -= s;
        float c[4];
        interpolate_cubic(f, c);
        ofs.push_back(s);
        for (int k = 0; k < 4; k++) coef.push_back((short)std::min(std::max(cvRoundf(c[k] * 2048.f), -32768), 32767));
    }
}

}  // namespace

struct uh_orb {
    uh_ctx* ctx = nullptr;
    // plan key
    int w = 0, h = 0, batch = 0;
    uh_feat_params fp{-1, 4000, 8, 1.2f, 0.f};
    bool planned = false;
    bool blur_first = true;        // ORBextractor::doGaussianBlur()
    bool nonmaxima = false;        // debug::Debug::isString("orb_nonmaxima") (ORBextractor.cpp:1146-1148)
    bool nm_attr = false;
    bool sel_attr = false;         // select_kernel's dynamic-LDS attribute has been set on this context's device
    int lvl_first = 0, lvl_end = -1;   // pyramid-level shard [first, end) this instance extracts (end < 0: all levels)
    int iniTh = 20, minTh = 7;     // precalculateParams resets these on every parameter change (:478-479)
    Plan plan;
    std::vector<CellDesc> cells;
    size_t frame_stride = 0, cand_stride = 0, sel_stride = 0;
    int lds_entries = 0;
    uh::DevBuf d_plan, d_cells, d_xofs, d_xcoef, d_yofs, d_ycoef;
    uh::DevBuf d_pyr, d_score, d_cand, d_work, d_sel, d_cell_counts, d_level_counts;
    uh::DevBuf d_tickets;   // select_kernel: workgroups of a (frame, level) that have finished their cells
    bool pair_ok[kMaxLevels] = {};  // levels l and l+1 can be built by one resize_pair_kernel launch (the footprints fit its staging area)
    bool pair_fusion = true;       // UH_ORB_PYRAMID=chain: one launch per level
    int pair_from = 4;             // three to six frames: levels >= pair_from are built two per launch (UH_ORB_PAIR_FROM for the A/B)
    bool pyr_fused = false;        // the whole pyramid in one launch (pyramid_fused_kernel): the plan's rectangles fit LDS (UH_ORB_PYRAMID=pair | chain for the launch-per-level forms)
    PyrPlan pyr_plan{};
    uh::DevBuf d_pyr_tiles;
    bool pyr_attr = false;
    bool fuse_fast = false;        // every cell fits cell_nms_kernel<true>'s staging buffers: no strength map, no fast_score launch
    int nms_tile_bytes = 0, nms_lds_bytes = 0;   // cell_nms_kernel<true>'s dynamic LDS: strength tile | patch / candidate lists
    bool score_valid = false;      // d_score holds the last extraction's strength maps (uh_orb_debug_level computes them on demand)
    // staging for the host-pointer API
    uh::DevBuf d_in, d_kps, d_desc, d_counts;
    uh::DevBuf d_early_ticket;     // und_early_kernel's workgroup counter (0 between launches)
    uh::MappedBuf h_early;         // uh_orb_extract_frame_dev_begin: [completion word | count | undistorted keypoints (position + octave)]
    unsigned long long early_seq = 0;
    struct Pending {               // between uh_orb_extract_frame_dev_begin and _end
        bool active = false, direct = false, und_direct = false;
        unsigned long long word = 0;
        uh_keypoint* kps = nullptr; uint8_t* desc = nullptr; float* und_xy = nullptr;
        int cap = 0;
        size_t o_cnt = 0, o_kps = 0, o_desc = 0, o_und = 0;
    } pending;
    uh::MappedBuf h_out;           // one-frame form: [completion word | count | keypoints | descriptors] when the caller's buffers are not pinned
    unsigned long long seq = 0;
    CamModel cam{};                // uh_orb_set_camera: on = 1 -> the one-frame entry can return the undistorted keypoints
    uh::DevBuf d_raw;              // an interleaved BGR / BGRA frame on its way to gray (uh_orb_extract_frame)
};

namespace {

// Mirrors ORBextractor::precalculateParams (:468-515), ComputePyramid sizes (:1369-1370) and the cell geometry of
// ComputeKeyPoints_thread (:899-976).
int make_plan(uh_orb* o, int w, int h, int batch) {
    const uh_feat_params& fp = o->fp;
    UH_REQUIRE(fp.nOctaveLevels >= 1 && fp.nOctaveLevels <= kMaxLevels, "orb: nOctaveLevels=%d outside [1,%d]", fp.nOctaveLevels, kMaxLevels);
    UH_REQUIRE(fp.scaleFactor > 1.0f, "orb: scaleFactor must be > 1");
    UH_REQUIRE(fp.maxFeatures >= 0, "orb: maxFeatures < 0");
    UH_REQUIRE(w >= 1 && h >= 1 && w <= kMaxDim && h <= kMaxDim, "orb: image %dx%d outside [1,%d]", w, h, kMaxDim);
    Plan& P = o->plan;
    std::memset(&P, 0, sizeof(P));
    const int nl = fp.nOctaveLevels;
    P.nlevels = nl;
    P.iniTh = o->iniTh;
    P.minTh = o->minTh;
    P.maxFeatures = fp.maxFeatures;
    std::vector<float> scale(nl, 1.f), inv(nl, 1.f);
    for (int i = 1; i < nl; i++) scale[i] = scale[i - 1] * fp.scaleFactor;
    for (int i = 0; i < nl; i++) inv[i] = 1.0f / scale[i];
    std::vector<int> nFeat(nl, 0);
    {
        const float factor = 1.0f / fp.scaleFactor;
        float nDesired = fp.maxFeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
        int sum = 0;
        for (int l = 0; l < nl - 1; l++) { nFeat[l] = cvRoundf(nDesired); sum += nFeat[l]; nDesired *= factor; }
        nFeat[nl - 1] = std::max(fp.maxFeatures - sum, 0);
    }
    // pyramid-level shard: the per-level budgets above are fixed by the FULL level count, so a level extracted
    // alone yields exactly the rows the full extraction holds for it; levels outside the shard get no tiles, cells or budget
    const int lvl_first = std::min(std::max(o->lvl_first, 0), nl);
    const int lvl_end = o->lvl_end < 0 ? nl : std::min(std::max(o->lvl_end, lvl_first), nl);
    P.lvl_end = lvl_end;
    o->cells.clear();
    std::vector<int> xofs, yofs;
    std::vector<short> xcoef, ycoef;
    size_t img_off = 0, cand_off = 0, sel_off = 0;
    int tile_begin = 0, cell_begin = 0;
    const float imageRatio = (float)w / h;   // :900 (level 0 is the input size)
    int max_level_need = 0;
    for (int l = 0; l < nl; l++) {
        LevelDesc& L = P.lv[l];
        L.w = cvRoundf((float)w * inv[l]);
        L.h = cvRoundf((float)h * inv[l]);
        UH_REQUIRE(L.w >= 1 && L.h >= 1, "orb: pyramid level %d is empty (%dx%d)", l, L.w, L.h);
        L.pitch = (L.w + 63) & ~63;
        L.img_off = (int)img_off;
        img_off += (size_t)L.pitch * L.h;
        L.scale = scale[l];
        L.scaledPatchSize = (int)(PATCH_SIZE * scale[l]);
        const bool mine = l >= lvl_first && l < lvl_end;
        L.nDesired = mine ? nFeat[l] : 0;
        L.tiles_x = mine ? uh_div_up(L.w, 64) : 0;
        L.tiles_y = mine ? uh_div_up(L.h, 16) : 0;
        L.tile_begin = tile_begin;
        tile_begin += L.tiles_x * L.tiles_y;
        L.xtap_off = (int)xofs.size();
        L.ytap_off = (int)yofs.size();
        if (l > 0) {
            cubic_taps(P.lv[l - 1].w, L.w, xofs, xcoef);
            cubic_taps(P.lv[l - 1].h, L.h, yofs, ycoef);
        }
        // cell grid
        const int levelCols = (int)std::sqrt((float)L.nDesired / (5 * imageRatio));
        const int levelRows = (int)(imageRatio * levelCols);
        L.cell_begin = cell_begin;
        L.nCells = 0;
        L.cellCap = 1;
        L.nfeaturesCell = 0;
        if (levelCols > 0 && levelRows > 0) {
            const int minB = EDGE, maxBX = L.w - EDGE, maxBY = L.h - EDGE;
            const int W = maxBX - minB, H = maxBY - minB;
            const int cellW = (int)std::ceil((float)W / levelCols);
            const int cellH = (int)std::ceil((float)H / levelRows);
            const int nCells = levelRows * levelCols;
            UH_REQUIRE(nCells <= kMaxCellsPerLevel, "orb: level %d needs %d cells (> %d)", l, nCells, kMaxCellsPerLevel);
            L.nCells = nCells;
            L.nfeaturesCell = (int)std::ceil((float)L.nDesired / nCells);
            std::vector<int> iniXCol(levelCols);
            float hY = cellH + 6;
            int cap = 1;
            for (int i = 0; i < levelRows; i++) {
                const float iniY = minB + i * cellH - 3;
                bool rowSkipped = false;
                if (i == levelRows - 1) { hY = maxBY + 3 - iniY; if (hY <= 0) rowSkipped = true; }
                float hX = cellW + 6;
                for (int j = 0; j < levelCols; j++) {
                    CellDesc C{0, 0, 0, 0, 0};
                    float iniX;
                    if (i == 0) { iniX = minB + j * cellW - 3; iniXCol[j] = (int)iniX; } else iniX = iniXCol[j];
                    bool skipped = rowSkipped;
                    if (!skipped && j == levelCols - 1) { hX = maxBX + 3 - iniX; if (hX <= 0) skipped = true; }
                    if (skipped) C.skipped = 1;
                    else {
                        const int y0 = (int)iniY, y1 = (int)(iniY + hY), x0 = (int)iniX, x1 = (int)(iniX + hX);
                        // a cell sub-image that leaves the level makes the reference's cv::Mat::rowRange/colRange throw
                        UH_REQUIRE(x0 >= 0 && y0 >= 0 && x1 <= L.w && y1 <= L.h,
                                   "orb: level %d cell (%d,%d) leaves the image (the reference throws cv::Exception here)", l, i, j);
                        C.x0 = (short)(x0 + 3); C.x1 = (short)std::max(x1 - 3, x0 + 3);
                        C.y0 = (short)(y0 + 3); C.y1 = (short)std::max(y1 - 3, y0 + 3);
                        const int iw = C.x1 - C.x0, ih = C.y1 - C.y0;
                        cap = std::max(cap, ((iw + 1) / 2) * ((ih + 1) / 2));
                    }
                    o->cells.push_back(C);
                }
            }
            L.cellCap = cap;
        }
        cell_begin += L.nCells;
        L.cand_off = (int)cand_off;
        L.work_cap = L.nCells * L.cellCap;
        cand_off += (size_t)L.work_cap;
        L.sel_off = (int)sel_off;
        sel_off += (size_t)std::max(L.nDesired, 1);
        max_level_need = std::max(max_level_need, L.work_cap);
    }
    P.total_tiles = tile_begin;
    P.total_cells = cell_begin;
    {   // resize_pair_kernel: do the rectangles a tile of level l+1 needs (of level l, and of level l-1 behind it) fit the staging area?
        const char* e = getenv("UH_ORB_PYRAMID");
        o->pair_fusion = !(e && std::string(e) == "chain");
        if (const char* pf = getenv("UH_ORB_PAIR_FROM")) o->pair_from = std::max(1, atoi(pf));
        auto clampi = [](int v, int lo, int hi) { return std::min(std::max(v, lo), hi); };
        for (int l = 0; l < kMaxLevels; l++) o->pair_ok[l] = false;
        for (int l = 1; l + 1 < nl; l++) {
            const LevelDesc& S0 = P.lv[l - 1]; const LevelDesc& A = P.lv[l]; const LevelDesc& B = P.lv[l + 1];
            const int* ax = xofs.data() + A.xtap_off; const int* ay = yofs.data() + A.ytap_off;
            const int* bx = xofs.data() + B.xtap_off; const int* by = yofs.data() + B.ytap_off;
            bool fits = true;
            for (int tx0 = 0; tx0 < B.w && fits; tx0 += 64) {
                const int xl = std::min(tx0 + 64, B.w) - 1;
                const int a0 = clampi(bx[tx0] - 1, 0, A.w - 1), a1 = clampi(bx[xl] + 2, 0, A.w - 1);
                const int s0 = clampi(ax[a0] - 1, 0, S0.w - 1), s1 = clampi(ax[a1] + 2, 0, S0.w - 1);
                fits = a1 - a0 + 1 <= 96 && s1 - s0 + 1 <= 144;
            }
            for (int ty0 = 0; ty0 < B.h && fits; ty0 += 16) {
                const int yl = std::min(ty0 + 16, B.h) - 1;
                const int a0 = clampi(by[ty0] - 1, 0, A.h - 1), a1 = clampi(by[yl] + 2, 0, A.h - 1);
                const int s0 = clampi(ay[a0] - 1, 0, S0.h - 1), s1 = clampi(ay[a1] + 2, 0, S0.h - 1);
                fits = a1 - a0 + 1 <= 40 && s1 - s0 + 1 <= 40;
            }
            o->pair_ok[l] = fits;
        }
    }
    {   // pyramid_fused_kernel: every level's share of a spatial tile (R) and what the tile must compute of it for the level above (C), top down
        const char* e = getenv("UH_ORB_PYRAMID");
        const bool want = !(e && (std::string(e) == "chain" || std::string(e) == "pair"));
        o->pyr_fused = false;
        const int nlv = P.lvl_end;
        auto clampi = [](int v, int lo, int hi) { return std::min(std::max(v, lo), hi); };
        // ~256 square tiles (one workgroup per compute unit for ONE frame: the halo makes a tile compute ~5x its share, which costs nothing
        // while the chip is otherwise idle; batches of three frames or more keep the launch-per-level forms, measured faster there)
        for (int attempt = 0; want && nlv >= 1 && attempt < 8 && !o->pyr_fused; attempt++) {
            const int ncu = o->ctx->num_cus > 0 ? o->ctx->num_cus : 256;
            const double side = std::sqrt((double)w * h / (double)ncu) / std::pow(1.4142, attempt);
            int ntx = std::max(1, (int)std::lround(w / std::max(side, 8.0))), nty = std::max(1, (int)std::lround(h / std::max(side, 8.0)));
            // (sixteen waves of <= 128 registers: ONE workgroup per compute unit — a 257th tile would be a second round of the whole launch)
            while (attempt == 0 && ntx * nty > ncu && ntx > 1) --ntx;
            if ((long long)ntx * nty > 8192) break;
            std::vector<PyrTile> tiles((size_t)ntx * nty);
            PyrPlan pl{};
            pl.ntiles = ntx * nty; pl.nlevels = nlv;
            long long szA = 16, szB = 16, szRaw = 16, szH16 = 16, szH32 = 16;
            bool ok = true;
            for (int ty = 0; ty < nty && ok; ty++)
                for (int tx = 0; tx < ntx && ok; tx++) {
                    PyrTile& T = tiles[(size_t)ty * ntx + tx];
                    int cx0 = 0, cy0 = 0, cx1 = 0, cy1 = 0;   // C of the level above (empty)
                    for (int l = nlv - 1; l >= 0; l--) {
                        const LevelDesc& L = P.lv[l];
                        const int rx0 = (int)((long long)tx * L.w / ntx), rx1 = (int)((long long)(tx + 1) * L.w / ntx);
                        const int ry0 = (int)((long long)ty * L.h / nty), ry1 = (int)((long long)(ty + 1) * L.h / nty);
                        int x0 = rx0, x1 = rx1, y0 = ry0, y1 = ry1;
                        const bool r_empty = rx1 <= rx0 || ry1 <= ry0;
                        if (r_empty) { x0 = y0 = 0; x1 = y1 = 0; }
                        if (l + 1 < nlv && cx1 > cx0 && cy1 > cy0) {   // what C_{l+1} reads of this level
                            const LevelDesc& U = P.lv[l + 1];
                            const int* ux = xofs.data() + U.xtap_off; const int* uy = yofs.data() + U.ytap_off;
                            const int fx0 = clampi(ux[cx0] - 1, 0, L.w - 1), fx1 = clampi(ux[cx1 - 1] + 2, 0, L.w - 1) + 1;
                            const int fy0 = clampi(uy[cy0] - 1, 0, L.h - 1), fy1 = clampi(uy[cy1 - 1] + 2, 0, L.h - 1) + 1;
                            if (x1 <= x0 || y1 <= y0) { x0 = fx0; x1 = fx1; y0 = fy0; y1 = fy1; }
                            else { x0 = std::min(x0, fx0); x1 = std::max(x1, fx1); y0 = std::min(y0, fy0); y1 = std::max(y1, fy1); }
                        }
                        T.r[l][0] = (short)(r_empty ? 0 : rx0); T.r[l][1] = (short)(r_empty ? 0 : ry0); T.r[l][2] = (short)(r_empty ? 0 : rx1); T.r[l][3] = (short)(r_empty ? 0 : ry1);
                        T.c[l][0] = (short)x0; T.c[l][1] = (short)y0; T.c[l][2] = (short)x1; T.c[l][3] = (short)y1;
                        cx0 = x0; cy0 = y0; cx1 = x1; cy1 = y1;
                        const long long cw = x1 - x0, ch = y1 - y0, pitch = (cw + 3) & ~3ll;
                        if (cw <= 0 || ch <= 0) continue;
                        if (cw + 6 > kPyrThreads) ok = false;   // (a thread per column)
                        if (l & 1) szB = std::max(szB, pitch * ch); else szA = std::max(szA, pitch * ch);
                        if (l == 0) { szRaw = std::max(szRaw, ((cw + 6 + 3) & ~3ll) * (ch + 6)); szH16 = std::max(szH16, (ch + 6) * pitch * 2); }
                        else {
                            const LevelDesc& Sv = P.lv[l - 1];
                            const int* ly = yofs.data() + L.ytap_off;
                            for (int by = 0; by < ch; by += kPyrBand) {
                                const int g0 = y0 + by, g1 = y0 + std::min<long long>(by + kPyrBand, ch) - 1;
                                const int s0 = clampi(ly[g0] - 1, 0, Sv.h - 1), s1 = clampi(ly[g1] + 2, 0, Sv.h - 1);
                                szH32 = std::max(szH32, (long long)(s1 - s0 + 1) * pitch * 4);
                            }
                        }
                    }
                }
            auto al16 = [](long long v) { return (int)((v + 15) & ~15ll); };
            pl.szA = al16(szA); pl.szB = al16(szB); pl.szRaw = al16(szRaw); pl.szH16 = al16(szH16); pl.szH32 = al16(szH32);
            const long long lds = (long long)pl.szA + std::max(pl.szB, pl.szRaw + pl.szH16) + pl.szH32;
            if (!ok || lds > 150 * 1024) continue;
            o->pyr_plan = pl;
            int rcp = o->d_pyr_tiles.reserve(tiles.size() * sizeof(PyrTile));
            if (rcp) return rcp;
            UH_HIP_CHECK(hipSetDevice(o->ctx->device));
            UH_HIP_CHECK(hipMemcpyAsync(o->d_pyr_tiles.p, tiles.data(), tiles.size() * sizeof(PyrTile), hipMemcpyHostToDevice, o->ctx->stream));
            UH_HIP_CHECK(hipStreamSynchronize(o->ctx->stream));
            o->pyr_fused = true;
        }
    }
    {   // the FAST strength is computed inside cell_nms_kernel when every cell fits its staging buffers (UH_ORB_FAST=map: the two-launch form)
        bool fits = true;
        for (const CellDesc& C : o->cells) {
            const int iw = C.x1 - C.x0, ih = C.y1 - C.y0;
            if (C.skipped || iw <= 0 || ih <= 0) continue;
            fits = fits && (iw + 2) * (ih + 2) <= kNmsTileBytes && (iw + 12) * (ih + 6) <= kNmsPatchBytes;
        }
        const char* e = getenv("UH_ORB_FAST");
        o->fuse_fast = fits && !(e && std::string(e) == "map");
        int most_px = 1, most_patch = 1, most_tile = 16;
        for (const CellDesc& C : o->cells) {
            const int iw = C.x1 - C.x0, ih = C.y1 - C.y0;
            if (C.skipped || iw <= 0 || ih <= 0) continue;
            most_px = std::max(most_px, iw * ih);
            most_patch = std::max(most_patch, (iw + 12) * (ih + 6));   // (dword-staged: up to 3 columns before and 3 after the patch)
            most_tile = std::max(most_tile, (iw + 2) * (ih + 2));   // (the tile's frame of zeros)
        }
        // candidate lists of the four waves: each <= half of its quarter of the raster (rounded up to 64) + 64
        const int list_bytes = 4 * ((((most_px + 3) / 4 + 63) & ~63) / 2 + 64) * 4;
        o->nms_tile_bytes = (most_tile + 15) & ~15;
        o->nms_lds_bytes = o->nms_tile_bytes + ((std::max(most_patch, list_bytes) + 15) & ~15);
    }
    // selection workgroups per level: one per 16 cells (a cell per wave) while the launch stays within one workgroup per CU; large
    // batches fall back towards one workgroup per level (every workgroup repeats the level's quota redistribution)
    {
        int want = 0;
        for (int l = 0; l < P.nlevels; l++) want += std::max(1, std::min(32, uh_div_up(std::max(P.lv[l].nCells, 1), kSelWaves)));
        const double f = std::min(1.0, 256.0 / ((double)std::max(want, 1) * std::max(batch, 1)));
        P.sel_wg_begin[0] = 0;
        for (int l = 0; l < kMaxLevels; l++) {
            int parts = 0;
            if (l < P.nlevels) parts = std::max(1, (int)(f * std::max(1, std::min(32, uh_div_up(std::max(P.lv[l].nCells, 1), kSelWaves)))));
            P.sel_wg_begin[l + 1] = P.sel_wg_begin[l] + parts;
        }
    }
    o->frame_stride = (img_off + 255) & ~(size_t)255;
    o->cand_stride = cand_off;
    o->sel_stride = sel_off;
    {   // dynamic LDS of select_kernel: a workgroup's filtered cell lists + partition scratch, and the level's concatenation in the level-wide
        // pass (a few times nDesired): 32 KiB for the usual budgets (four workgroups fit a CU), up to 96 KiB for large ones
        int most = 1;
        for (int l = 0; l < P.nlevels; l++) most = std::max(most, P.lv[l].nDesired);
        o->lds_entries = std::min(24576, std::max(8192, ((6 * most + 2047) / 2048) * 2048));
    }
    int rc;
    UH_HIP_CHECK(hipSetDevice(o->ctx->device));   // first: the attribute below belongs to the context's device, not to whatever the calling thread had current
    if (!o->sel_attr) {   // once per object
        UH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 24576 * 4));   // (the largest plan's need)
        o->sel_attr = true;
    }
    hipStream_t st = o->ctx->stream;
    auto up = [&](uh::DevBuf& b, const void* src, size_t bytes) -> int {
        int r = b.reserve(std::max(bytes, (size_t)16));
        if (r) return r;
        if (bytes) UH_HIP_CHECK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st));
        return UH_OK;
    };
    if ((rc = up(o->d_plan, &P, sizeof(P)))) return rc;
    if ((rc = up(o->d_cells, o->cells.data(), o->cells.size() * sizeof(CellDesc)))) return rc;
    if ((rc = up(o->d_xofs, xofs.data(), xofs.size() * sizeof(int)))) return rc;
    if ((rc = up(o->d_xcoef, xcoef.data(), xcoef.size() * sizeof(short)))) return rc;
    if ((rc = up(o->d_yofs, yofs.data(), yofs.size() * sizeof(int)))) return rc;
    if ((rc = up(o->d_ycoef, ycoef.data(), ycoef.size() * sizeof(short)))) return rc;
    UH_HIP_CHECK(hipStreamSynchronize(st));   // the host vectors die here
    if ((rc = o->d_pyr.reserve(o->frame_stride * batch))) return rc;
    if ((rc = o->d_score.reserve(o->frame_stride * batch))) return rc;
    if ((rc = o->d_cand.reserve(std::max<size_t>(o->cand_stride, 1) * batch * 4))) return rc;
    if ((rc = o->d_work.reserve(std::max<size_t>(o->cand_stride, 1) * batch * 8))) return rc;
    if ((rc = o->d_sel.reserve(std::max<size_t>(o->sel_stride, 1) * batch * 4))) return rc;
    if ((rc = o->d_cell_counts.reserve(std::max<size_t>(P.total_cells, 1) * batch * 8))) return rc;
    if ((rc = o->d_level_counts.reserve((size_t)kMaxLevels * batch * 4))) return rc;
    {   // select_kernel's ticket counters: zero between launches (the last workgroup of a level puts its counter back)
        const size_t tb = (size_t)kMaxLevels * batch * 4;
        const bool fresh = o->d_tickets.cap < tb;
        if ((rc = o->d_tickets.reserve(tb))) return rc;
        if (fresh) UH_HIP_CHECK(hipMemsetAsync(o->d_tickets.p, 0, o->d_tickets.cap, o->ctx->stream));
    }
    o->w = w; o->h = h; o->batch = batch;
    o->planned = true;
    return UH_OK;
}

int run_frames(uh_orb* o, const uint8_t* d_imgs, int w, int h, size_t stride, size_t img_frame_stride, int batch,
               KeyPointOut* d_kps, uint8_t* d_desc, int cap_per_frame, int* d_counts, float* d_und = nullptr, uh_dev_frame* fr = nullptr, const EarlyUnd* early = nullptr) {
    int rc;
    if (!o->planned || o->w != w || o->h != h || o->batch < batch) {
        if ((rc = make_plan(o, w, h, std::max(batch, o->planned && o->w == w && o->h == h ? o->batch : 0)))) return rc;
    }
    UH_HIP_CHECK(hipSetDevice(o->ctx->device));
    hipStream_t st = o->ctx->stream;
    const Plan& P = o->plan;
    uint8_t* pyr = o->d_pyr.as<uint8_t>();
    const LevelDesc& L0 = P.lv[0];
    if (P.lvl_end == 0 || P.total_tiles == 0) {   // empty shard (more ranks than levels): no keypoints
        UH_HIP_CHECK(hipMemsetAsync(d_counts, 0, sizeof(int) * batch, st));
        return UH_OK;
    }
    const bool fused = o->pyr_fused && batch == 1;
    if (fused) {   // blur + every resize in one launch
        const PyrPlan& pl = o->pyr_plan;
        const size_t lds = (size_t)pl.szA + (size_t)std::max(pl.szB, pl.szRaw + pl.szH16) + (size_t)pl.szH32;
        if (!o->pyr_attr) {
            UH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pyramid_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            o->pyr_attr = true;
        }
        const PyrTaps tp{o->d_xofs.as<int>(), o->d_xcoef.as<short>(), o->d_yofs.as<int>(), o->d_ycoef.as<short>()};
        UH_LAUNCH(o->ctx, pyramid_fused_kernel, dim3(pl.ntiles, 1, batch), dim3(kPyrThreads), lds, d_imgs, w, h, stride, img_frame_stride, pyr, o->frame_stride, P, pl,
                  (const PyrTile*)o->d_pyr_tiles.as<PyrTile>(), tp, o->blur_first ? 1 : 0);
    } else if (o->blur_first) {
        UH_LAUNCH(o->ctx,blur7_kernel, dim3(uh_div_up(w, 64), uh_div_up(h, 16), batch), dim3(256), 0, d_imgs, w, h, stride,
                           img_frame_stride, pyr + L0.img_off, L0.pitch, o->frame_stride);
    } else {
        UH_LAUNCH(o->ctx,copy_kernel, dim3(uh_div_up(w, 256), h, batch), dim3(256), 0, d_imgs, w, h, stride,
                           img_frame_stride, pyr + L0.img_off, L0.pitch, o->frame_stride);
    }
    for (int l = 1; l < P.lvl_end && !fused; l++) {
        const LevelDesc& S = P.lv[l - 1];
        const LevelDesc& D = P.lv[l];
        // two levels per launch: for one or two frames everywhere (pure launch latency); up to 16 frames six only above level 3, where the
        // rebuilt rectangles are a few dozen tiles and a saved launch (~9.5 us) is worth more than their recomputation
        if (o->pair_fusion && (batch <= 2 || (batch <= 6 && l >= o->pair_from)) && l + 1 < P.lvl_end && o->pair_ok[l]) {
            const LevelDesc& E = P.lv[l + 1];
            const LevelTaps ta{o->d_xofs.as<int>() + D.xtap_off, o->d_xcoef.as<short>() + (size_t)D.xtap_off * 4,
                               o->d_yofs.as<int>() + D.ytap_off, o->d_ycoef.as<short>() + (size_t)D.ytap_off * 4};
            const LevelTaps tb{o->d_xofs.as<int>() + E.xtap_off, o->d_xcoef.as<short>() + (size_t)E.xtap_off * 4,
                               o->d_yofs.as<int>() + E.ytap_off, o->d_ycoef.as<short>() + (size_t)E.ytap_off * 4};
            const int tilesA = uh_div_up(D.w, 64) * uh_div_up(D.h, 16), tilesB = uh_div_up(E.w, 64) * uh_div_up(E.h, 16);
            UH_LAUNCH(o->ctx, resize_pair_kernel, dim3(tilesA + tilesB, 1, batch), dim3(256), 0, (const uint8_t*)pyr, o->frame_stride, S, D, E, ta, tb, tilesA);
            ++l;
            continue;
        }
        UH_LAUNCH(o->ctx,resize_cubic_kernel, dim3(uh_div_up(D.w, 64), uh_div_up(D.h, 16), batch), dim3(256), 0,
                           pyr + S.img_off, S.w, S.h, S.pitch, pyr + D.img_off, D.w, D.h, D.pitch, o->frame_stride,
                           o->d_xofs.as<int>() + D.xtap_off, o->d_xcoef.as<short>() + (size_t)D.xtap_off * 4,
                           o->d_yofs.as<int>() + D.ytap_off, o->d_ycoef.as<short>() + (size_t)D.ytap_off * 4);
    }
    o->score_valid = !o->fuse_fast;
    if (!o->fuse_fast) {
        UH_LAUNCH(o->ctx,fast_score_kernel, dim3(P.total_tiles, batch), dim3(256), 0, P, pyr, o->d_score.as<uint8_t>(),
                           o->frame_stride);
    }
    if (P.total_cells > 0) {
        if (o->fuse_fast) {
            UH_LAUNCH(o->ctx,cell_nms_kernel<true>, dim3(batch, P.total_cells), dim3(256), (size_t)o->nms_lds_bytes, P, o->d_cells.as<CellDesc>(),
                               (const uint8_t*)pyr, o->frame_stride, o->d_cand.as<uint32_t>(), o->cand_stride, o->d_cell_counts.as<int>(),
                               o->nms_tile_bytes);
        } else {
            UH_LAUNCH(o->ctx,cell_nms_kernel<false>, dim3(batch, P.total_cells), dim3(256), 0, P, o->d_cells.as<CellDesc>(),
                               (const uint8_t*)o->d_score.as<uint8_t>(), o->frame_stride, o->d_cand.as<uint32_t>(), o->cand_stride,
                               o->d_cell_counts.as<int>(), 0);
        }
    }
    UH_LAUNCH(o->ctx,select_kernel, dim3(P.sel_wg_begin[P.nlevels], batch), dim3(kSelThreads), (size_t)o->lds_entries * 4, P,
                       o->d_cells.as<CellDesc>(), o->d_cand.as<uint32_t>(), o->cand_stride, o->d_cell_counts.as<int>(),
                       o->d_work.as<uint32_t>(), o->cand_stride * 2, o->d_sel.as<uint32_t>(), o->sel_stride,
                       o->d_level_counts.as<int>(), o->lds_entries, o->d_tickets.as<int>());
    if (o->nonmaxima) {
        const size_t lds = (size_t)std::max(P.maxFeatures, 1) * 5 + 64;
        UH_REQUIRE(lds <= 150 * 1024, "orb: orb_nonmaxima with %d features per level does not fit LDS", P.maxFeatures);
        if (!o->nm_attr) {
            UH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(nonmax_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            o->nm_attr = true;
        }
        UH_LAUNCH(o->ctx,nonmax_kernel, dim3(P.nlevels, batch), dim3(64), lds, P, o->d_sel.as<uint32_t>(), o->sel_stride, o->d_level_counts.as<int>());
    }
    if (early)   // (one frame: the undistorted keypoints to the host ahead of the descriptors)
        UH_LAUNCH(o->ctx, und_early_kernel, dim3(kEarlyGroups), dim3(kEarlyThreads), 0, P, (const uint32_t*)o->d_sel.as<uint32_t>(), (const int*)o->d_level_counts.as<int>(), o->cam, *early);
    const int slots = std::min(std::max(P.maxFeatures, 1), std::max(cap_per_frame, 1));
    UH_LAUNCH(o->ctx,describe_kernel, dim3(uh_div_up(slots, 4) * batch), dim3(256), 0, P, pyr, o->frame_stride,
                       o->d_sel.as<uint32_t>(), o->sel_stride, o->d_level_counts.as<int>(), d_kps, d_desc, cap_per_frame,
                       d_counts, o->nonmaxima ? 1 : -1, o->cam, o->cam.on ? d_und : nullptr, batch, fr ? fr->desc() : (uint8_t*)nullptr,
                       fr ? fr->kd_in() : (float4*)nullptr);
    UH_HIP_CHECK(hipGetLastError());
    return UH_OK;
}

}  // namespace

extern "C" {

int uh_orb_create(uh_ctx* ctx, uh_orb** out) {
    UH_REQUIRE(ctx && out, "uh_orb_create: NULL argument");
    uh_orb* o = new uh_orb();
    o->ctx = ctx;
    *out = o;
    return UH_OK;
}

void uh_orb_destroy(uh_orb* o) { delete o; }

int uh_orb_set_params(uh_orb* o, const uh_feat_params* fp) {
    UH_REQUIRE(o && fp, "uh_orb_set_params: NULL argument");
    // Feature2DSerializable::FeatParams::operator== ignores sensitivity (feature2dserializable.h:48)
    const bool same = o->planned && o->fp.nthreads == fp->nthreads && o->fp.maxFeatures == fp->maxFeatures &&
                      o->fp.nOctaveLevels == fp->nOctaveLevels && o->fp.scaleFactor == fp->scaleFactor;
    o->fp = *fp;
    if (!same) { o->planned = false; o->iniTh = 20; o->minTh = 7; }   // precalculateParams (:478-479)
    return UH_OK;
}

// ---- Feature2DSerializable::toStream / fromStream (feature2dserializable.cpp:76-113) for the ORB extractor: u64 signature 1828374733,
// u64 type tag (F2D_ORB = 0), the parameter string as u32 length + bytes (io_utils: toStream__(std::string)), then
// ORBextractor::toStream_impl (ORBextractor.cpp:417-419): the raw 20-byte FeatParams.
int uh_orb_to_stream(const uh_orb* o, const char* str_params, uint8_t* out, uint64_t cap, uint64_t* size) {
    UH_REQUIRE(o && size, "uh_orb_to_stream: NULL argument");
    const size_t ls = str_params ? std::strlen(str_params) : 0;
    *size = 8 + 8 + 4 + ls + sizeof(uh_feat_params);
    if (!out) return UH_OK;
    if (cap < *size) { uh::set_error("uh_orb_to_stream: %llu bytes needed, capacity %llu", (unsigned long long)*size, (unsigned long long)cap); return UH_ECAPACITY; }
    const uint64_t sig = 1828374733ull, type = 0;
    const uint32_t l32 = (uint32_t)ls;
    uint8_t* w = out;
    std::memcpy(w, &sig, 8); w += 8;
    std::memcpy(w, &type, 8); w += 8;
    std::memcpy(w, &l32, 4); w += 4;
    if (ls) { std::memcpy(w, str_params, ls); w += ls; }
    std::memcpy(w, &o->fp, sizeof(uh_feat_params));
    return UH_OK;
}

// Reads such a stream into an existing extractor object.  str_params_out (may be NULL) receives the parameter string, NUL-terminated,
// truncated to str_cap - 1 characters; *consumed (may be NULL) the number of bytes read.
int uh_orb_from_stream(uh_orb* o, const uint8_t* data, uint64_t nbytes, char* str_params_out, uint64_t str_cap, uint64_t* consumed) {
    UH_REQUIRE(o && data, "uh_orb_from_stream: NULL argument");
    UH_REQUIRE(nbytes >= 20, "uh_orb_from_stream: stream too short");
    uint64_t sig, type;
    uint32_t ls;
    std::memcpy(&sig, data, 8); std::memcpy(&type, data + 8, 8); std::memcpy(&ls, data + 16, 4);
    UH_REQUIRE(sig == 1828374733ull, "Feature2DSerializable::fromStream signature error in stream");   // feature2dserializable.cpp:91-92
    // F2D_ORB = 0; the GridExtractor types (AKAZE / BRISK / ORB-grid / FREAK / SURF) are out of scope
    UH_REQUIRE(type == 0, "uh_orb_from_stream: extractor type %llu is not F2D_ORB (grid extractors are not supported)", (unsigned long long)type);
    UH_REQUIRE((uint64_t)20 + ls + sizeof(uh_feat_params) <= nbytes, "uh_orb_from_stream: truncated stream");
    if (str_params_out && str_cap) {
        const size_t n = std::min<size_t>(ls, (size_t)str_cap - 1);
        std::memcpy(str_params_out, data + 20, n);
        str_params_out[n] = 0;
    }
    uh_feat_params fp;
    std::memcpy(&fp, data + 20 + ls, sizeof(fp));
    if (consumed) *consumed = 20 + ls + sizeof(fp);
    return uh_orb_set_params(o, &fp);
}

int uh_orb_get_params(const uh_orb* o, uh_feat_params* fp) {
    UH_REQUIRE(o && fp, "uh_orb_get_params: NULL argument");
    *fp = o->fp;
    return UH_OK;
}

int uh_orb_set_blur(uh_orb* o, int do_blur) {
    UH_REQUIRE(o, "uh_orb_set_blur: NULL");
    o->blur_first = do_blur != 0;
    return UH_OK;
}

// the reference's debug switch "orb_nonmaxima" (debug::Debug::addString, ORBextractor.cpp:1146-1148): radius-3 suppression per level
int uh_orb_set_nonmaxima(uh_orb* o, int on) {
    UH_REQUIRE(o, "uh_orb_set_nonmaxima: NULL");
    o->nonmaxima = on != 0;
    return UH_OK;
}

// ORBextractor::setSensitivity (:457-466)
int uh_orb_set_sensitivity(uh_orb* o, float v) {
    UH_REQUIRE(o, "uh_orb_set_sensitivity: NULL");
    if (v > 1) v = 1;
    if (v <= 0) v = 0;
    o->fp.sensitivity = v;
    v = 1 - v;
    o->iniTh = (int)(v * 10 + 10);
    o->minTh = (int)(v * 4 + 3);
    o->planned = false;
    return UH_OK;
}

// Pyramid-level shard for multi-GPU extraction of ONE frame: this instance extracts levels [first, end) only
// (end < 0: up to the last level).  Level l depends on level l-1 (ORBextractor.cpp:1379), so the chain is built redundantly
// up to end-1; keypoints and descriptors of the shards concatenated in level order equal the full extraction.
int uh_orb_set_level_range(uh_orb* o, int first, int end) {
    UH_REQUIRE(o, "uh_orb_set_level_range: NULL");
    UH_REQUIRE(first >= 0 && (end < 0 || end >= first), "uh_orb_set_level_range: bad range [%d,%d)", first, end);
    if (o->lvl_first != first || o->lvl_end != end) o->planned = false;
    o->lvl_first = first;
    o->lvl_end = end;
    return UH_OK;
}

int uh_orb_max_keypoints(const uh_orb* o) { return o ? std::max(o->fp.maxFeatures, 0) : 0; }

int uh_orb_extract_dev(uh_orb* o, const uint8_t* d_imgs, int w, int h, size_t stride, size_t frame_stride, int batch,
                       uh_keypoint* d_kps, uint8_t* d_desc, int cap_per_frame, int32_t* d_counts) {
    UH_REQUIRE(o, "uh_orb_extract_dev: NULL extractor");
    UH_REQUIRE(batch >= 1, "uh_orb_extract_dev: batch < 1");
    UH_REQUIRE(d_imgs && d_kps && d_desc && d_counts, "uh_orb_extract_dev: NULL buffer");
    UH_REQUIRE(stride >= (size_t)w && cap_per_frame >= 1, "uh_orb_extract_dev: bad stride/capacity");
    return run_frames(o, d_imgs, w, h, stride, frame_stride, batch, reinterpret_cast<KeyPointOut*>(d_kps), d_desc,
                      cap_per_frame, d_counts);
}

static int extract_finish(uh_orb* o, int* n_out);

// early != NULL: return as soon as the undistorted keypoints are on the host (*early = this object's array of them); uh_orb_extract_frame_dev_end completes the call
static int extract_one(uh_orb* o, const uint8_t* img, int w, int h, size_t stride, int cn, uh_keypoint* kps, uint8_t* desc, float* und_xy, int cap,
                       int* n_out, uh_dev_frame* fr = nullptr, const uh_keypoint** early = nullptr) {
    UH_REQUIRE(o && n_out, "uh_orb_extract: NULL argument");
    UH_REQUIRE(!o->pending.active, "uh_orb_extract: uh_orb_extract_frame_dev_begin without its _end");
    *n_out = 0;
    if (early) *early = nullptr;
    if (img == nullptr || w <= 0 || h <= 0) {   // ORBextractor.cpp:1254 — empty image: silent return (a device frame becomes the empty frame)
        if (fr) { int rc0 = uh::dev_frame_reserve(fr, 1); if (rc0 || (!fr->host_tree && (rc0 = uh::kd_build_launch(fr, nullptr, 0, 0, 0)))) return rc0; }
        return UH_OK;
    }
    UH_REQUIRE(cn == 1 || cn == 3 || cn == 4, "uh_orb_extract_frame: %d channels (1 = gray, 3 = BGR, 4 = BGRA)", cn);
    UH_REQUIRE(stride >= (size_t)w * cn, "uh_orb_extract: stride < row bytes");
    UH_REQUIRE(cap >= 0 && (cap == 0 || (kps && desc)), "uh_orb_extract: NULL output buffer");
    UH_REQUIRE(!und_xy || o->cam.on, "uh_orb_extract_frame: undistorted keypoints asked for but no camera set (uh_orb_set_camera)");
    const int maxk = std::max(o->fp.maxFeatures, 1);
    int rc;
    if (fr) {   // the frame also stays on the device (descriptors, undistorted keypoints, kd-tree): uh_orb_extract_frame_dev
        UH_REQUIRE(fr->ctx == o->ctx, "uh_orb_extract_frame_dev: the device frame belongs to another context");
        UH_REQUIRE(o->cam.on, "uh_orb_extract_frame_dev: no camera set (uh_orb_set_camera): Frame::und_kpts need one");
        UH_REQUIRE(o->lvl_first == 0 && o->lvl_end < 0, "uh_orb_extract_frame_dev: a pyramid-level shard is not a frame");
        if ((rc = uh::dev_frame_reserve(fr, maxk))) return rc;
    }
    UH_HIP_CHECK(hipSetDevice(o->ctx->device));
    hipStream_t st = o->ctx->stream;
    // One frame is latency, not bandwidth: no copy engine and no stream synchronisation on the way.  A frame in pinned memory
    // (uh_host_alloc / hipHostMalloc / hipHostRegister) is fetched by a kernel with 16-byte loads; keypoints, descriptors and the
    // count are written by the last kernel into pinned memory — the caller's own buffers if they are pinned, this object's block
    // otherwise — and a completion word follows them, which the host polls.
    if ((rc = o->d_in.reserve((size_t)w * h + 16))) return rc;
    // (the 16-byte loads / stores of the pinned paths need 16-byte aligned bases: an interior pointer — an ROI, a numpy view inside a pinned
    // block — that is not aligned takes the staging paths below instead of relying on the hardware's unaligned-access mode)
    const uint8_t* h_img = static_cast<const uint8_t*>(uh::device_alias_of_host(img));
    if (cn != 1) {   // colour frame: to gray on the way in (cv::cvtColor COLOR_BGR2GRAY, frameextractor.cpp:2960) — from pinned memory in place, else through a staged copy
        const uint8_t* src = h_img;
        size_t sstride = stride;
        if (!src) {
            if ((rc = o->d_raw.reserve((size_t)w * cn * h + 16))) return rc;
            UH_HIP_CHECK(hipMemcpy2DAsync(o->d_raw.p, (size_t)w * cn, img, stride, (size_t)w * cn, h, hipMemcpyHostToDevice, st));
            src = o->d_raw.as<uint8_t>(); sstride = (size_t)w * cn;
        }
        UH_LAUNCH(o->ctx, ingest_bgr_kernel, dim3(uh_div_up(((w + 3) / 4) * h, 256)), dim3(256), 0, src, w, h, sstride, cn, o->d_in.as<uint8_t>());
    } else {
    if (h_img && stride == (size_t)w && (reinterpret_cast<uintptr_t>(h_img) & 15) != 0) h_img = nullptr;   // (the strided path copies 4-byte pieces of any alignment)
    if (h_img) {
        const int chunks = stride == (size_t)w ? (int)(((size_t)w * h + 15) / 16) : ((w + 15) / 16) * h;
        UH_LAUNCH(o->ctx, ingest_kernel, dim3(uh_div_up(chunks, 256)), dim3(256), 0, h_img, w, h, stride, o->d_in.as<uint8_t>());
    } else if (stride == (size_t)w) {   // pageable frame: through the runtime's staging copy
        UH_HIP_CHECK(hipMemcpyAsync(o->d_in.p, img, (size_t)w * h, hipMemcpyHostToDevice, st));
    } else {
        UH_HIP_CHECK(hipMemcpy2DAsync(o->d_in.p, w, img, stride, w, h, hipMemcpyHostToDevice, st));
    }
    }
    const uint8_t* d_img = o->d_in.as<uint8_t>();
    const size_t in_stride = (size_t)w;
    const int slots = std::min(maxk, std::max(cap, 1));
    const size_t o_cnt = 64, o_kps = 128, o_desc = o_kps + (((size_t)maxk * sizeof(uh_keypoint) + 63) & ~(size_t)63), o_und = o_desc + (size_t)maxk * 32,
                 total = o_und + ((und_xy || fr) ? (size_t)maxk * 8 : 0);
    if ((rc = o->h_out.reserve(total))) return rc;
    char* hb = o->h_out.host<char>();
    char* db = o->h_out.dev<char>();
    KeyPointOut* d_kps = cap > 0 ? static_cast<KeyPointOut*>(uh::device_alias_of_host(kps)) : nullptr;
    uint8_t* d_desc = cap > 0 ? static_cast<uint8_t*>(uh::device_alias_of_host(desc)) : nullptr;
    const bool direct = d_kps && d_desc && (reinterpret_cast<uintptr_t>(d_desc) & 15) == 0;   // (describe_kernel stores descriptors 16 bytes wide)
    if (!direct) { d_kps = reinterpret_cast<KeyPointOut*>(db + o_kps); d_desc = reinterpret_cast<uint8_t*>(db + o_desc); }
    // (the undistorted positions: into the caller's array if it is pinned and 8-byte aligned, else into this object's block)
    float* d_und = und_xy ? static_cast<float*>(uh::device_alias_of_host(und_xy)) : nullptr;
    // (with staged keypoints the launch runs at cap_per_frame = maxk, so describe_kernel stores up to maxk positions — the caller's
    // array holds only `cap` of them: write into it only when the launch itself is bounded by `cap`, or `cap` covers every slot)
    const bool und_direct = d_und && (reinterpret_cast<uintptr_t>(d_und) & 7) == 0 && (direct || cap >= maxk);
    if ((und_xy || fr) && !und_direct) d_und = reinterpret_cast<float*>(db + o_und);
    const int cap_launch = direct ? slots : maxk;
    EarlyUnd ea{};
    if (early) {
        if ((rc = o->h_early.reserve(128 + (size_t)maxk * sizeof(uh_keypoint) + 64))) return rc;
        char* de = o->h_early.dev<char>();
        ea.kpts = reinterpret_cast<KeyPointOut*>(de + 128); ea.count = reinterpret_cast<int*>(de + 64); ea.word_ptr = reinterpret_cast<unsigned long long*>(de);
        ea.word = ++o->early_seq; ea.cap = cap_launch;
        if (!o->d_early_ticket.p) {
            if ((rc = o->d_early_ticket.reserve(64))) return rc;
            UH_HIP_CHECK(hipMemsetAsync(o->d_early_ticket.p, 0, 64, st));
        }
        ea.ticket = o->d_early_ticket.as<unsigned>();
    }
    rc = run_frames(o, d_img, w, h, in_stride, (size_t)in_stride * h, 1, d_kps, d_desc, cap_launch, reinterpret_cast<int*>(db + o_cnt), d_und, fr, early ? &ea : nullptr);
    if (rc) return rc;
    // the completion word as its own one-thread launch behind the last kernel (a ticket counter inside describe_kernel — one system-scope
    // release and one same-address atomic per workgroup — cost 10 us more than this launch)
    const unsigned long long word = ++o->seq;
    if ((rc = uh::post_host_word(o->ctx, reinterpret_cast<unsigned long long*>(db), word))) return rc;
    // Frame::create_kdtree (frameextractor.cpp:4258) as one more launch BEHIND the completion word: the host gets its keypoints and goes on
    // while the tree is built; the projection matcher adopts it with uh_projmatch_set_frame_dev (no D2H -> host build -> H2D)
    if (fr && !fr->host_tree && (rc = uh::kd_build_launch(fr, o->d_level_counts.as<int>(), o->plan.nlevels, cap_launch, 0))) return rc;
    uh_orb::Pending& pd = o->pending;
    pd.active = true; pd.direct = direct; pd.und_direct = und_direct; pd.word = word; pd.kps = kps; pd.desc = desc; pd.und_xy = und_xy; pd.cap = cap;
    pd.o_cnt = o_cnt; pd.o_kps = o_kps; pd.o_desc = o_desc; pd.o_und = o_und;
    if (early) {   // the keypoints' undistorted positions and octaves are enough for the caller to go on with (the kd-tree): the rest is collected by _end
        char* he = o->h_early.host<char>();
        if ((rc = uh::wait_host_word(reinterpret_cast<volatile unsigned long long*>(he), ea.word, st, "uh_orb_extract_frame_dev_begin"))) { pd.active = false; return rc; }
        const int n = *reinterpret_cast<const int*>(he + 64);
        *n_out = n;
        *early = reinterpret_cast<const uh_keypoint*>(he + 128);
        if (n > cap) { pd.active = false; (void)hipStreamSynchronize(st); uh::set_error("uh_orb_extract: %d keypoints but capacity %d", n, cap); return UH_ECAPACITY; }
        return UH_OK;
    }
    return extract_finish(o, n_out);
}

static int extract_finish(uh_orb* o, int* n_out) {
    uh_orb::Pending& pd = o->pending;
    UH_REQUIRE(pd.active, "uh_orb_extract_frame_dev_end: no extraction in flight");
    pd.active = false;
    char* hb = o->h_out.host<char>();
    int rc;
    if ((rc = uh::wait_host_word(reinterpret_cast<volatile unsigned long long*>(hb), pd.word, o->ctx->stream, "uh_orb_extract"))) return rc;
    const int n = *reinterpret_cast<const int*>(hb + pd.o_cnt);
    if (n_out) *n_out = n;
    if (n > pd.cap) { uh::set_error("uh_orb_extract: %d keypoints but capacity %d", n, pd.cap); return UH_ECAPACITY; }
    if (!pd.direct && n > 0) {
        std::memcpy(pd.kps, hb + pd.o_kps, (size_t)n * sizeof(uh_keypoint));
        std::memcpy(pd.desc, hb + pd.o_desc, (size_t)n * 32);
    }
    if (pd.und_xy && !pd.und_direct && n > 0) std::memcpy(pd.und_xy, hb + pd.o_und, (size_t)n * 8);
    return UH_OK;
}

int uh_orb_extract(uh_orb* o, const uint8_t* img, int w, int h, size_t stride, uh_keypoint* kps, uint8_t* desc, int cap, int* n_out) {
    return extract_one(o, img, w, h, stride, 1, kps, desc, nullptr, cap, n_out);
}
int uh_orb_extract_frame(uh_orb* o, const uint8_t* img, int w, int h, size_t stride, int channels, uh_keypoint* kps, uint8_t* desc, float* und_xy,
                         int cap, int* n_out) {
    return extract_one(o, img, w, h, stride, channels, kps, desc, und_xy, cap, n_out);
}
int uh_orb_extract_frame_dev(uh_orb* o, const uint8_t* img, int w, int h, size_t stride, int channels, uh_keypoint* kps, uint8_t* desc, float* und_xy,
                             int cap, int* n_out, uh_dev_frame* frame) {
    UH_REQUIRE(frame, "uh_orb_extract_frame_dev: NULL device frame");
    return extract_one(o, img, w, h, stride, channels, kps, desc, und_xy, cap, n_out, frame);
}
// uh_orb_extract_frame_dev in two halves: _begin returns as soon as the undistorted keypoints (position + octave; the other fields are not
// filled in yet) are on the host — *und_kpts_early points at this object's array of them, valid until the next extraction — so that the
// caller builds the kd-tree (uh_projmatch_set_frame_dev with a host-built tree) while describe_kernel still runs; _end waits for the
// descriptors / keypoints / und_xy the _begin call was given buffers for.  Nothing else may be extracted in between.
int uh_orb_extract_frame_dev_begin(uh_orb* o, const uint8_t* img, int w, int h, size_t stride, int channels, uh_keypoint* kps, uint8_t* desc, float* und_xy,
                                   int cap, int* n_out, uh_dev_frame* frame, const uh_keypoint** und_kpts_early) {
    UH_REQUIRE(frame && und_kpts_early, "uh_orb_extract_frame_dev_begin: NULL argument");
    return extract_one(o, img, w, h, stride, channels, kps, desc, und_xy, cap, n_out, frame, und_kpts_early);
}
int uh_orb_extract_frame_dev_end(uh_orb* o, int* n_out) {
    UH_REQUIRE(o, "uh_orb_extract_frame_dev_end: NULL extractor");
    if (!o->pending.active) return UH_OK;   // (an empty image: _begin had nothing to launch)
    return extract_finish(o, n_out);
}
int uh_orb_set_camera(uh_orb* o, const uh_camera* cam) {
    UH_REQUIRE(o, "uh_orb_set_camera: NULL extractor");
    CamModel c{};
    if (cam) {
        UH_REQUIRE(cam->n_dist >= 0 && cam->n_dist <= 8 && cam->fx != 0.f && cam->fy != 0.f, "uh_orb_set_camera: %d distortion coefficients / zero focal length", cam->n_dist);
        c.fx = cam->fx; c.fy = cam->fy; c.cx = cam->cx; c.cy = cam->cy; c.ifx = 1.0 / c.fx; c.ify = 1.0 / c.fy;
        for (int i = 0; i < cam->n_dist; i++) c.k[i] = cam->dist[i];
        c.ffx = cam->fx; c.ffy = cam->fy; c.fcx = cam->cx; c.fcy = cam->cy;
        c.on = 1;
    }
    o->cam = c;
    return UH_OK;
}
// host-only form of the same arithmetic (a caller that has keypoints from elsewhere; tests): n points in, n out
int uh_undistort_points_host(const uh_camera* cam, const float* xy, int n, float* out_xy) {
    UH_REQUIRE(cam && (n == 0 || (xy && out_xy)) && cam->n_dist >= 0 && cam->n_dist <= 8, "uh_undistort_points_host: bad argument");
    CamModel c{};
    c.fx = cam->fx; c.fy = cam->fy; c.cx = cam->cx; c.cy = cam->cy; c.ifx = 1.0 / c.fx; c.ify = 1.0 / c.fy;
    for (int i = 0; i < cam->n_dist; i++) c.k[i] = cam->dist[i];
    c.ffx = cam->fx; c.ffy = cam->fy; c.fcx = cam->cx; c.fcy = cam->cy;
    for (int i = 0; i < n; i++) undistort_point(c, xy[2 * i], xy[2 * i + 1], out_xy[2 * i], out_xy[2 * i + 1]);
    return UH_OK;
}

// detectAndCompute for a BATCH of frames, host memory in and out (the tracker's call shape when frames arrive in groups, and what a
// C++ host without device buffers of its own uses): one H2D copy of the frames, the batched launch set, one D2H copy each of the
// fixed-capacity keypoint / descriptor blocks and the counts, one synchronisation.  Frame f's rows are kps + f * cap, desc + f * cap * 32,
// the first counts[f] of them valid.  Pinned buffers (uh_host_alloc) make the copies asynchronous DMA.
int uh_orb_extract_batch(uh_orb* o, const uint8_t* imgs, int w, int h, size_t stride, size_t frame_stride, int batch, uh_keypoint* kps, uint8_t* desc,
                         int cap, int32_t* counts) {
    UH_REQUIRE(o && imgs && kps && desc && counts, "uh_orb_extract_batch: NULL argument");
    UH_REQUIRE(w > 0 && h > 0 && batch >= 1 && stride >= (size_t)w && frame_stride >= stride * (size_t)(h - 1) + (size_t)w, "uh_orb_extract_batch: bad geometry");
    const int maxk = std::max(o->fp.maxFeatures, 1);
    UH_REQUIRE(cap >= maxk, "uh_orb_extract_batch: capacity %d below maxFeatures %d", cap, maxk);
    int rc;
    UH_HIP_CHECK(hipSetDevice(o->ctx->device));
    hipStream_t st = o->ctx->stream;
    if ((rc = o->d_in.reserve((size_t)w * h * batch))) return rc;
    if ((rc = o->d_kps.reserve((size_t)cap * batch * sizeof(uh_keypoint)))) return rc;
    if ((rc = o->d_desc.reserve((size_t)cap * batch * 32))) return rc;
    if ((rc = o->d_counts.reserve((size_t)batch * 4 + 16))) return rc;
    if (stride == (size_t)w && frame_stride == (size_t)w * h) UH_HIP_CHECK(hipMemcpyAsync(o->d_in.p, imgs, (size_t)w * h * batch, hipMemcpyHostToDevice, st));
    else for (int f = 0; f < batch; f++)
        UH_HIP_CHECK(hipMemcpy2DAsync(o->d_in.as<uint8_t>() + (size_t)f * w * h, w, imgs + (size_t)f * frame_stride, stride, w, h, hipMemcpyHostToDevice, st));
    rc = run_frames(o, o->d_in.as<uint8_t>(), w, h, w, (size_t)w * h, batch, o->d_kps.as<KeyPointOut>(), o->d_desc.as<uint8_t>(), cap, o->d_counts.as<int>());
    if (rc) return rc;
    UH_HIP_CHECK(hipMemcpyAsync(kps, o->d_kps.p, (size_t)cap * batch * sizeof(uh_keypoint), hipMemcpyDeviceToHost, st));
    UH_HIP_CHECK(hipMemcpyAsync(desc, o->d_desc.p, (size_t)cap * batch * 32, hipMemcpyDeviceToHost, st));
    UH_HIP_CHECK(hipMemcpyAsync(counts, o->d_counts.p, (size_t)batch * 4, hipMemcpyDeviceToHost, st));
    UH_HIP_CHECK(hipStreamSynchronize(st));
    return UH_OK;
}

// Debug/verification taps (tests compare every stage with the oracle): copies one level of one frame's pyramid or
// score map to the host.  which: 0 = pyramid, 1 = FAST strength map.
int uh_orb_debug_level(uh_orb* o, int frame, int level, int which, uint8_t* out, int* w_out, int* h_out) {
    UH_REQUIRE(o && o->planned, "uh_orb_debug_level: extractor has not run yet");
    UH_REQUIRE(frame >= 0 && frame < o->batch && level >= 0 && level < o->plan.nlevels, "uh_orb_debug_level: bad frame/level");
    const LevelDesc& L = o->plan.lv[level];
    if (w_out) *w_out = L.w;
    if (h_out) *h_out = L.h;
    if (!out) return UH_OK;
    if (which && !o->score_valid) {   // the fused extraction keeps no strength map: compute it from the pyramid of the last extraction
        UH_HIP_CHECK(hipSetDevice(o->ctx->device));
        UH_LAUNCH(o->ctx, fast_score_kernel, dim3(o->plan.total_tiles, o->batch), dim3(256), 0, o->plan, (const uint8_t*)o->d_pyr.as<uint8_t>(),
                  o->d_score.as<uint8_t>(), o->frame_stride);
        o->score_valid = true;
    }
    const uint8_t* base = (which ? o->d_score.as<uint8_t>() : o->d_pyr.as<uint8_t>()) + (size_t)frame * o->frame_stride + L.img_off;
    UH_HIP_CHECK(hipMemcpy2DAsync(out, L.w, base, L.pitch, L.w, L.h, hipMemcpyDeviceToHost, o->ctx->stream));
    UH_HIP_CHECK(hipStreamSynchronize(o->ctx->stream));
    return UH_OK;
}

}  // extern "C"
