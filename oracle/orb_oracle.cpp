// TEST INFRASTRUCTURE ONLY — CPU oracle for the multi-scale ORB extractor.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
//
// PARITY UNPINNED: the reference extractor (src/featureextractors/ORBextractor.cpp) does its pixel arithmetic
// inside OpenCV, which is neither vendored under /root/reference nor installed here, and the reference holds no
// golden vectors for this stage.  This file restates ORBextractor.cpp line by line and, for the seven OpenCV
// primitives it leans on, the published OpenCV 4.x (>= 4.3) algorithms, chosen and documented in DESIGN.md:
//   GaussianBlur 8U  : bit-exact fixed-point path (8.8 kernel with error diffusion, round-half-up at 16 bits)
//   resize CUBIC 8U  : fixed-point reference path (A=-0.75, 11-bit coefficients, (v + 2^21) >> 22)
//   FAST_t<16>, cornerScore<16>, KeyPointsFilter::retainBest (nth_element at n-1), fastAtan2, cvRound
// Float math is compiled un-contracted (-ffp-contract=off), as the reference's effective flags are.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/ucoslam_hip_orb_pattern.inc"

namespace {

struct KeyPoint {          // layout of cv::KeyPoint (7 x 4 bytes), ORBextractor.cpp:1297 memcpy's these
    float x, y, size, angle, response;
    int32_t octave, class_id;
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> d;
    Image() {}
    Image(int W, int H) : w(W), h(H), d((size_t)W * H) {}
    const uint8_t* row(int y) const { return d.data() + (size_t)y * w; }
    uint8_t* row(int y) { return d.data() + (size_t)y * w; }
};

constexpr int PATCH_SIZE = 31, HALF_PATCH_SIZE = 15, EDGE_THRESHOLD = 19;  // ORBextractor.cpp:74-76
const int8_t kPattern[1024] = {UH_ORB_PATTERN_VALUES};

inline int cvRound(float v) { return (int)lrintf(v); }   // round-half-even (default FP environment)
inline int cvRound(double v) { return (int)lrint(v); }
inline int cvFloor(double v) { int i = (int)v; return i - (i > v); }
inline int cvCeil(double v) { int i = (int)v; return i + (i < v); }
inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// ---------------------------------------------------------------- cv::GaussianBlur(8U, 7x7, sigma 2, REFLECT_101)
// OpenCV >= 4.3: getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED (8 fractional bits, symmetric
// error diffusion, centre takes the remainder so the taps sum to exactly 256), then the fixed-point separable
// filter: horizontal 8.8 accumulate, vertical 16.16 accumulate, (v + 2^15) >> 16.
void gaussian_kernel_fixed(int n, double sigma, int out[]) {
    std::vector<double> k(n);
    double sum = 0;
    double scale2x = -0.5 * 0.25 / (sigma * sigma);
    for (int i = 0, x = 1 - n; i < n; i++, x += 2) { k[i] = std::exp((double)(x * x) * scale2x); sum += k[i]; }
    sum = 1.0 / sum;
    for (int i = 0; i < n; i++) k[i] *= sum;
    double err = 0;
    long total = 0;
    for (int i = 0; i < n / 2; i++) {
        double adj = k[i] * 256.0 + err;
        long v0 = lrint(adj);
        err = adj - (double)v0;
        out[i] = out[n - 1 - i] = (int)v0;
        total += v0;
    }
    out[n / 2] = (int)(256 - 2 * total);
}

void gaussian_blur7(const uint8_t* src, int w, int h, size_t stride, Image& dst) {
    int k[7];
    gaussian_kernel_fixed(7, 2.0, k);   // = {18,34,48,56,48,34,18}
    dst = Image(w, h);
    std::vector<uint32_t> tmp((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* r = src + (size_t)y * stride;
        uint32_t* o = &tmp[(size_t)y * w];
        for (int x = 0; x < w; x++) {
            uint32_t s = 0;
            if (x >= 3 && x + 3 < w) for (int i = 0; i < 7; i++) s += (uint32_t)k[i] * r[x + i - 3];     // interior: no reflection
            else for (int i = 0; i < 7; i++) s += (uint32_t)k[i] * r[reflect101(x + i - 3, w)];
            o[x] = s;
        }
    }
    for (int y = 0; y < h; y++) {
        const uint32_t* rows[7];
        for (int j = 0; j < 7; j++) rows[j] = &tmp[(size_t)reflect101(y + j - 3, h) * w];
        uint8_t* o = dst.row(y);
        for (int x = 0; x < w; x++) {
            uint32_t s = 0;
            for (int j = 0; j < 7; j++) s += (uint32_t)k[j] * rows[j][x];
            o[x] = (uint8_t)((s + 32768u) >> 16);
        }
    }
}

// ---------------------------------------------------------------- cv::resize(INTER_CUBIC) on 8UC1
inline void interpolate_cubic(float x, float* c) {
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}
inline short sat_short(float v) { int i = cvRound(v); return (short)std::min(std::max(i, -32768), 32767); }

// tap tables: ofs[d] = floor of the source coordinate, coef[4d..4d+3] = round(w*2048)
void cubic_taps(int ssize, int dsize, std::vector<int>& ofs, std::vector<short>& coef) {
    double inv_scale = (double)dsize / ssize;
    double scale = 1.0 / inv_scale;
    ofs.resize(dsize);
    coef.resize((size_t)dsize * 4);
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cvFloor(f);
        f -= s;
        float c[4];
        interpolate_cubic(f, c);
        ofs[d] = s;
        for (int k = 0; k < 4; k++) coef[(size_t)d * 4 + k] = sat_short(c[k] * 2048.f);
    }
}

void resize_cubic(const Image& src, Image& dst, int dw, int dh) {
    dst = Image(dw, dh);
    std::vector<int> xo, yo;
    std::vector<short> xa, ya;
    cubic_taps(src.w, dw, xo, xa);
    cubic_taps(src.h, dh, yo, ya);
    std::vector<int> rows((size_t)src.h * dw);   // horizontal pass of every source row
    std::vector<int> sxk((size_t)dw * 4);
    for (int x = 0; x < dw; x++) for (int k = 0; k < 4; k++) sxk[(size_t)x * 4 + k] = std::min(std::max(xo[x] - 1 + k, 0), src.w - 1);   // taps clamp at the ROI edge
    for (int y = 0; y < src.h; y++) {
        const uint8_t* S = src.row(y);
        int* o = &rows[(size_t)y * dw];
        for (int x = 0; x < dw; x++) {
            const int* sx = &sxk[(size_t)x * 4];
            const short* a = &xa[(size_t)x * 4];
            o[x] = S[sx[0]] * a[0] + S[sx[1]] * a[1] + S[sx[2]] * a[2] + S[sx[3]] * a[3];
        }
    }
    for (int y = 0; y < dh; y++) {
        const int* r[4];
        for (int k = 0; k < 4; k++) r[k] = &rows[(size_t)std::min(std::max(yo[y] - 1 + k, 0), src.h - 1) * dw];
        const short* b = &ya[(size_t)y * 4];
        uint8_t* o = dst.row(y);
        for (int x = 0; x < dw; x++) {
            const int acc = r[0][x] * b[0] + r[1][x] * b[1] + r[2][x] * b[2] + r[3][x] * b[3];
            const int v = (acc + (1 << 21)) >> 22;
            o[x] = (uint8_t)std::min(std::max(v, 0), 255);
        }
    }
}

// ---------------------------------------------------------------- cv::FAST(img, kps, thr, nonmax=true), TYPE_9_16
const int kCircle[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                            {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

int corner_score16(const uint8_t* ptr, const int pixel[25], int threshold) {
    const int N = 25;
    int v = ptr[0];
    short d[N];
    for (int k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, (int)d[k + 4]);
        a = std::min(a, (int)d[k + 5]);
        a = std::min(a, (int)d[k + 6]);
        a = std::min(a, (int)d[k + 7]);
        a = std::min(a, (int)d[k + 8]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        b = std::max(b, (int)d[k + 3]);
        b = std::max(b, (int)d[k + 4]);
        b = std::max(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, (int)d[k + 6]);
        b = std::max(b, (int)d[k + 7]);
        b = std::max(b, (int)d[k + 8]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

// The sub-image is rows [0,rows) x cols [0,cols) of `img` (row stride `step`).  Keypoints in sub-image coords,
// raster order, KeyPoint(x, y, 7, -1, score).
void fast9_16(const uint8_t* img, int cols, int rows, int step, int threshold, std::vector<KeyPoint>& out) {
    out.clear();
    if (rows < 7 || cols < 7) return;
    const int K = 8, N = 25;
    int pixel[25];
    for (int k = 0; k < 16; k++) pixel[k] = kCircle[k][0] + kCircle[k][1] * step;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
    threshold = std::min(std::max(threshold, 0), 255);
    std::vector<uint8_t> bufmem((size_t)cols * 3, 0);
    std::vector<int> cpmem((size_t)(cols + 1) * 3, 0);
    uint8_t* buf[3] = {bufmem.data(), bufmem.data() + cols, bufmem.data() + 2 * cols};
    int* cpbuf[3] = {cpmem.data() + 1, cpmem.data() + 1 + (cols + 1), cpmem.data() + 1 + 2 * (cols + 1)};
    for (int i = 3; i < rows - 2; i++) {
        const uint8_t* ptr = img + (size_t)i * step + 3;
        uint8_t* curr = buf[(i - 3) % 3];
        int* cornerpos = cpbuf[(i - 3) % 3];
        std::memset(curr, 0, cols);
        int ncorners = 0;
        if (i < rows - 3) {
            for (int j = 3; j < cols - 3; j++, ptr++) {
                int v = ptr[0];
                int vt_lo = v - threshold, vt_hi = v + threshold;
                // cv::FAST's pruning: a 9-arc contains one end of every diameter, so all eight diameters must have an end
                // that is darker (bit 0) resp. brighter (bit 1) than the centre by more than the threshold
                auto cls = [&](int k) { const int x = ptr[pixel[k]]; return x < vt_lo ? 1 : (x > vt_hi ? 2 : 0); };
                int dmask = cls(0) | cls(8);
                if (dmask == 0) continue;
                dmask &= cls(2) | cls(10);
                dmask &= cls(4) | cls(12);
                dmask &= cls(6) | cls(14);
                if (dmask == 0) continue;
                dmask &= cls(1) | cls(9);
                dmask &= cls(3) | cls(11);
                dmask &= cls(5) | cls(13);
                dmask &= cls(7) | cls(15);
                bool found = false;
                int count = 0;
                if (dmask & 1)
                    for (int k = 0; k < N && !found; k++) {       // darker arc
                        if (ptr[pixel[k]] < vt_lo) { if (++count > K) found = true; } else count = 0;
                    }
                count = 0;
                if (dmask & 2)
                    for (int k = 0; k < N && !found; k++) {       // brighter arc
                        if (ptr[pixel[k]] > vt_hi) { if (++count > K) found = true; } else count = 0;
                    }
                if (found) {
                    cornerpos[ncorners++] = j;
                    curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;
        const uint8_t* prev = buf[(i - 4 + 3) % 3];
        const uint8_t* pprev = buf[(i - 5 + 3) % 3];
        cornerpos = cpbuf[(i - 4 + 3) % 3];
        ncorners = cornerpos[-1];
        for (int k = 0; k < ncorners; k++) {
            int j = cornerpos[k];
            int score = prev[j];
            if (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
                score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1]) {
                out.push_back(KeyPoint{(float)j, (float)(i - 1), 7.f, -1.f, (float)score, 0, -1});
            }
        }
    }
}

// ---------------------------------------------------------------- cv::KeyPointsFilter::retainBest (OpenCV 4.x)
void retain_best(std::vector<KeyPoint>& kps, int n_points) {
    if (n_points >= 0 && kps.size() > (size_t)n_points) {
        if (n_points == 0) { kps.clear(); return; }
        auto greater = [](const KeyPoint& a, const KeyPoint& b) { return a.response > b.response; };
        std::nth_element(kps.begin(), kps.begin() + n_points - 1, kps.end(), greater);
        float ambiguous = kps[n_points - 1].response;
        auto new_end = std::partition(kps.begin() + n_points, kps.end(),
                                      [ambiguous](const KeyPoint& k) { return k.response >= ambiguous; });
        kps.resize(new_end - kps.begin());
    }
}

// ---------------------------------------------------------------- cv::fastAtan2 (scalar atan_f32, degrees)
const float atan2_p1 = 0.9997878412794807f * (float)(180 / M_PI);
const float atan2_p3 = -0.3258083974640975f * (float)(180 / M_PI);
const float atan2_p5 = 0.1555786518463281f * (float)(180 / M_PI);
const float atan2_p7 = -0.04432655554792128f * (float)(180 / M_PI);
float fast_atan2(float y, float x) {
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// ---------------------------------------------------------------- the extractor (ORBextractor.cpp)
struct Extractor {
    int nlevels = 8, maxFeatures = 4000;
    float scaleFactor = 1.2f;
    int iniThFAST = 20, minThFAST = 7;
    bool blurFirst = true;                       // ORBextractor.h:120 _doGaussianBlurAtFirst
    std::vector<float> scale, invScale;
    std::vector<int> nFeat;
    std::vector<int> umax;
    std::vector<Image> pyr;                      // un-padded levels: the reference's 19-px border is never read
                                                 // (cells span [16,dim-16), patches stay >= 1 px inside the ROI)

    Extractor() {                                // ORBextractor.cpp:426-451
        umax.resize(HALF_PATCH_SIZE + 1);
        int v, v0, vmax = cvFloor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
        int vmin = cvCeil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
        const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
        for (v = 0; v <= vmax; ++v) umax[v] = cvRound(std::sqrt(hp2 - v * v));
        for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
    }

    void precalc() {                             // ORBextractor.cpp:468-515
        scale.assign(nlevels, 1.f);
        invScale.assign(nlevels, 1.f);
        for (int i = 1; i < nlevels; i++) scale[i] = scale[i - 1] * scaleFactor;
        for (int i = 0; i < nlevels; i++) invScale[i] = 1.0f / scale[i];
        nFeat.assign(nlevels, 0);
        float factor = 1.0f / scaleFactor;
        float nDesired = maxFeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
        int sum = 0;
        for (int l = 0; l < nlevels - 1; l++) {
            nFeat[l] = cvRound(nDesired);
            sum += nFeat[l];
            nDesired *= factor;
        }
        nFeat[nlevels - 1] = std::max(maxFeatures - sum, 0);
    }

    void pyramid(const Image& in) {              // ORBextractor.cpp:1355-1393
        pyr.resize(nlevels);
        for (int l = 0; l < nlevels; l++) {
            float s = invScale[l];
            int w = cvRound((float)in.w * s), h = cvRound((float)in.h * s);
            if (l == 0) pyr[0] = in;
            else resize_cubic(pyr[l - 1], pyr[l], w, h);
        }
    }

    float ic_angle(const Image& im, int cx, int cy) const {   // ORBextractor.cpp:79-106
        int m_01 = 0, m_10 = 0;
        const uint8_t* center = im.row(cy) + cx;
        for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
        int step = im.w;
        for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
            int v_sum = 0, d = umax[v];
            for (int u = -d; u <= d; ++u) {
                int vp = center[u + v * step], vm = center[u - v * step];
                v_sum += (vp - vm);
                m_10 += u * (vp + vm);
            }
            m_01 += v * v_sum;
        }
        return fast_atan2((float)m_01, (float)m_10);
    }

    mutable bool threw = false;   // the reference would have thrown a cv::Exception (cell outside the level image)
    bool nonMaximaSuppression = false;   // ORBextractor.h:189, switched on by the debug string "orb_nonmaxima"

    void keypoints_level(int level, std::vector<KeyPoint>& keypoints) const {   // ORBextractor.cpp:899-1078
        const Image& im = pyr[level];
        keypoints.clear();
        float imageRatio = (float)pyr[0].w / pyr[0].h;
        const int nDesiredFeatures = nFeat[level];
        const int levelCols = std::sqrt((float)nDesiredFeatures / (5 * imageRatio));
        const int levelRows = imageRatio * levelCols;
        const int minBorderX = EDGE_THRESHOLD, minBorderY = minBorderX;
        const int maxBorderX = im.w - EDGE_THRESHOLD, maxBorderY = im.h - EDGE_THRESHOLD;
        const int W = maxBorderX - minBorderX, H = maxBorderY - minBorderY;
        if (levelCols <= 0 || levelRows <= 0) return;   // the reference divides by zero here; defined as "no keypoints"
        const int cellW = std::ceil((float)W / levelCols);
        const int cellH = std::ceil((float)H / levelRows);
        const int nCells = levelRows * levelCols;
        const int nfeaturesCell = std::ceil((float)nDesiredFeatures / nCells);

        std::vector<std::vector<KeyPoint>> cellKps(nCells);
        std::vector<int> nToRetain(nCells, 0), nTotal(nCells, 0);
        std::vector<char> bNoMore(nCells, 0);
        std::vector<int> iniXCol(levelCols), iniYRow(levelRows);
        int nNoMore = 0, nToDistribute = 0;
        float hY = cellH + 6;
        for (int i = 0; i < levelRows; i++) {
            const float iniY = minBorderY + i * cellH - 3;
            iniYRow[i] = iniY;
            if (i == levelRows - 1) {
                hY = maxBorderY + 3 - iniY;
                if (hY <= 0) continue;
            }
            float hX = cellW + 6;
            for (int j = 0; j < levelCols; j++) {
                float iniX;
                if (i == 0) { iniX = minBorderX + j * cellW - 3; iniXCol[j] = iniX; }
                else iniX = iniXCol[j];
                if (j == levelCols - 1) {
                    hX = maxBorderX + 3 - iniX;
                    if (hX <= 0) continue;
                }
                // cellImage = rowRange(iniY, iniY+hY).colRange(iniX, iniX+hX); a range that leaves the image makes
                // the reference throw (cv::Mat's range assertion -> cv::Exception out of detectAndCompute): very flat or
                // very small levels whose rounded-up cell height overshoots.  Reported, not clipped.
                int y0 = (int)iniY, y1 = (int)(iniY + hY), x0 = (int)iniX, x1 = (int)(iniX + hX);
                if (x0 < 0 || y0 < 0 || x1 > im.w || y1 > im.h) { threw = true; keypoints.clear(); return; }
                int c = i * levelCols + j;
                if (y1 - y0 > 0 && x1 - x0 > 0) {
                    fast9_16(im.row(y0) + x0, x1 - x0, y1 - y0, im.w, iniThFAST, cellKps[c]);
                    if (cellKps[c].size() <= 3) fast9_16(im.row(y0) + x0, x1 - x0, y1 - y0, im.w, minThFAST, cellKps[c]);
                }
                const int nKeys = (int)cellKps[c].size();
                nTotal[c] = nKeys;
                if (nKeys > nfeaturesCell) { nToRetain[c] = nfeaturesCell; bNoMore[c] = 0; }
                else { nToRetain[c] = nKeys; nToDistribute += nfeaturesCell - nKeys; bNoMore[c] = 1; nNoMore++; }
            }
        }
        while (nToDistribute > 0 && nNoMore < nCells) {
            int nNewFeaturesCell = nfeaturesCell + std::ceil((float)nToDistribute / (nCells - nNoMore));
            nToDistribute = 0;
            for (int c = 0; c < nCells; c++) {
                if (bNoMore[c]) continue;
                if (nTotal[c] > nNewFeaturesCell) { nToRetain[c] = nNewFeaturesCell; bNoMore[c] = 0; }
                else { nToRetain[c] = nTotal[c]; nToDistribute += nNewFeaturesCell - nTotal[c]; bNoMore[c] = 1; nNoMore++; }
            }
        }
        const int scaledPatchSize = PATCH_SIZE * scale[level];
        for (int i = 0; i < levelRows; i++)
            for (int j = 0; j < levelCols; j++) {
                std::vector<KeyPoint>& keysCell = cellKps[i * levelCols + j];
                int nr = nToRetain[i * levelCols + j];
                retain_best(keysCell, nr);
                if ((int)keysCell.size() > nr) keysCell.resize(nr);
                for (KeyPoint& k : keysCell) {
                    k.x += iniXCol[j];
                    k.y += iniYRow[i];
                    k.octave = level;
                    k.size = scaledPatchSize;
                    keypoints.push_back(k);
                }
            }
        if ((int)keypoints.size() > nDesiredFeatures) {
            retain_best(keypoints, nDesiredFeatures);
            keypoints.resize(nDesiredFeatures);
        }
        for (KeyPoint& k : keypoints) k.angle = ic_angle(im, cvRound(k.x), cvRound(k.y));   // :1074, :516-523
    }

    void descriptor(const Image& im, const KeyPoint& kpt, uint8_t* desc) const {   // ORBextractor.cpp:113-153
        const float factorPI = (float)(M_PI / 180.f);
        float angle = (float)kpt.angle * factorPI;
        float a = (float)std::cos(angle), b = (float)std::sin(angle);
        const uint8_t* center = im.row(cvRound(kpt.y)) + cvRound(kpt.x);
        const int step = im.w;
        const int8_t* p = kPattern;
        for (int i = 0; i < 32; ++i) {
            int val = 0;
            for (int t = 0; t < 8; ++t, p += 4) {
                int t0 = center[cvRound(p[0] * b + p[1] * a) * step + cvRound(p[0] * a - p[1] * b)];
                int t1 = center[cvRound(p[2] * b + p[3] * a) * step + cvRound(p[2] * a - p[3] * b)];
                val |= (t0 < t1) << t;
            }
            desc[i] = (uint8_t)val;
        }
    }

    static constexpr int kRefThrows = -2147483647;   // extract(): the reference throws for this geometry

    // ORBextractor.cpp:1247-1353 (compute) + :1155-1230 (processLevel); returns total keypoints
    int extract(const uint8_t* img, int w, int h, size_t stride, std::vector<KeyPoint>& kps, std::vector<uint8_t>& desc) {
        kps.clear();
        desc.clear();
        if (w <= 0 || h <= 0) return 0;
        precalc();
        Image in;
        if (blurFirst) gaussian_blur7(img, w, h, stride, in);
        else { in = Image(w, h); for (int y = 0; y < h; y++) std::memcpy(in.row(y), img + (size_t)y * stride, w); }
        pyramid(in);
        for (int l = 0; l < nlevels; l++) {
            std::vector<KeyPoint> lk;
            keypoints_level(l, lk);
            if (threw) { kps.clear(); desc.clear(); return kRefThrows; }
            if (nonMaximaSuppression) {   // processLevel :1176-1205 (debug string "orb_nonmaxima")
                // kdtree.radiusSearch(res, keypoints, keypoints[i], 3) = every keypoint with squared distance < 9 (picoflann's
                // radius search returns exactly the brute-force disc, see proj_oracle.cpp / test_projmatch_oracle.py; the
                // order inside `res` is irrelevant for a maximum and for marking all smaller ones)
                for (KeyPoint& k : lk) k.class_id = 1;
                for (size_t i = 0; i < lk.size(); i++) {
                    if (!lk[i].class_id) continue;
                    int maxResponse = 0;
                    std::vector<size_t> res;
                    for (size_t j = 0; j < lk.size(); j++) {
                        const double dx = lk[i].x - lk[j].x, dy = lk[i].y - lk[j].y;
                        if (dx * dx + dy * dy < 9.0) res.push_back(j);
                    }
                    for (size_t j : res) if (lk[j].response > maxResponse) maxResponse = lk[j].response;
                    for (size_t j : res) if (lk[j].response < maxResponse) lk[j].class_id = 0;
                }
                lk.erase(std::remove_if(lk.begin(), lk.end(), [](const KeyPoint& k) { return k.class_id == 0; }), lk.end());
            }
            const Image& im = pyr[l];
            int maxX = im.w - 19, maxY = im.h - 19;   // computeDescriptors :1120-1137
            lk.erase(std::remove_if(lk.begin(), lk.end(), [&](const KeyPoint& k) {
                         return k.x < 19 || k.y < 19 || k.x > maxX || k.y > maxY; }), lk.end());
            size_t base = desc.size();
            desc.resize(base + lk.size() * 32);
            for (size_t i = 0; i < lk.size(); i++) descriptor(im, lk[i], desc.data() + base + i * 32);
            float sc = scale[l];
            if (l != 0) for (KeyPoint& k : lk) { k.x = (k.x + 0.5f) * sc; k.y = (k.y + 0.5f) * sc; }   // :1228-1229
            kps.insert(kps.end(), lk.begin(), lk.end());
        }
        return (int)kps.size();
    }
};

}  // namespace

extern "C" {

// Full extractor. kp_out: cap x 28 bytes, desc_out: cap x 32 bytes. Returns n (or -needed if cap too small, or -2147483647
// where the reference throws a cv::Exception: a FAST cell outside its level image).
int oracle_orb_extract(const uint8_t* img, int w, int h, size_t stride, int maxFeatures, int nlevels, float scaleFactor,
                       int blurFirst, void* kp_out, uint8_t* desc_out, int cap) {
    Extractor e;
    e.maxFeatures = maxFeatures;
    e.nlevels = nlevels;
    e.scaleFactor = scaleFactor;
    e.blurFirst = blurFirst != 0;
    std::vector<KeyPoint> kps;
    std::vector<uint8_t> desc;
    int n = e.extract(img, w, h, stride, kps, desc);
    if (n < 0) return n;
    if (n > cap) return -n;
    if (n) { std::memcpy(kp_out, kps.data(), (size_t)n * sizeof(KeyPoint)); std::memcpy(desc_out, desc.data(), (size_t)n * 32); }
    return n;
}

// same with the "orb_nonmaxima" switch (radius-3 suppression per level, kept keypoints carry class_id 1)
int oracle_orb_extract_nonmaxima(const uint8_t* img, int w, int h, size_t stride, int maxFeatures, int nlevels, float scaleFactor,
                                 int blurFirst, void* kp_out, uint8_t* desc_out, int cap) {
    Extractor e;
    e.maxFeatures = maxFeatures;
    e.nlevels = nlevels;
    e.scaleFactor = scaleFactor;
    e.blurFirst = blurFirst != 0;
    e.nonMaximaSuppression = true;
    std::vector<KeyPoint> kps;
    std::vector<uint8_t> desc;
    int n = e.extract(img, w, h, stride, kps, desc);
    if (n < 0) return n;
    if (n > cap) return -n;
    if (n) { std::memcpy(kp_out, kps.data(), (size_t)n * sizeof(KeyPoint)); std::memcpy(desc_out, desc.data(), (size_t)n * 32); }
    return n;
}

// Stage probes for known-answer tests -----------------------------------------------------------------------
void oracle_orb_gauss_kernel(int* out7) { gaussian_kernel_fixed(7, 2.0, out7); }
void oracle_orb_blur(const uint8_t* img, int w, int h, size_t stride, uint8_t* out) {
    Image d;
    gaussian_blur7(img, w, h, stride, d);
    std::memcpy(out, d.d.data(), d.d.size());
}
void oracle_orb_resize_cubic(const uint8_t* img, int w, int h, uint8_t* out, int dw, int dh) {
    Image s(w, h), d;
    std::memcpy(s.d.data(), img, (size_t)w * h);
    resize_cubic(s, d, dw, dh);
    std::memcpy(out, d.d.data(), d.d.size());
}
void oracle_orb_cubic_taps(int ssize, int dsize, int* ofs, short* coef) {
    std::vector<int> o;
    std::vector<short> c;
    cubic_taps(ssize, dsize, o, c);
    std::memcpy(ofs, o.data(), o.size() * sizeof(int));
    std::memcpy(coef, c.data(), c.size() * sizeof(short));
}
// level sizes and per-level feature budgets (ORBextractor.cpp:468-515, :1369-1370)
void oracle_orb_level_plan(int w, int h, int maxFeatures, int nlevels, float scaleFactor, int* lw, int* lh, int* nfeat,
                           float* scales) {
    Extractor e;
    e.maxFeatures = maxFeatures; e.nlevels = nlevels; e.scaleFactor = scaleFactor;
    e.precalc();
    for (int l = 0; l < nlevels; l++) {
        lw[l] = cvRound((float)w * e.invScale[l]);
        lh[l] = cvRound((float)h * e.invScale[l]);
        nfeat[l] = e.nFeat[l];
        scales[l] = e.scale[l];
    }
}
// pyramid level l of the (optionally blurred) image, un-padded
int oracle_orb_pyramid_level(const uint8_t* img, int w, int h, size_t stride, int nlevels, float scaleFactor, int blurFirst,
                             int level, uint8_t* out, int* ow, int* oh) {
    Extractor e;
    e.nlevels = nlevels; e.scaleFactor = scaleFactor; e.blurFirst = blurFirst != 0;
    e.precalc();
    Image in;
    if (e.blurFirst) gaussian_blur7(img, w, h, stride, in);
    else { in = Image(w, h); for (int y = 0; y < h; y++) std::memcpy(in.row(y), img + (size_t)y * stride, w); }
    e.pyramid(in);
    if (level < 0 || level >= nlevels) return -1;
    *ow = e.pyr[level].w; *oh = e.pyr[level].h;
    if (out) std::memcpy(out, e.pyr[level].d.data(), e.pyr[level].d.size());
    return 0;
}
// cv::FAST on a whole (sub-)image: returns n, writes (x,y,score) int triples
int oracle_fast_detect(const uint8_t* img, int cols, int rows, int step, int threshold, int* xys, int cap) {
    std::vector<KeyPoint> k;
    fast9_16(img, cols, rows, step, threshold, k);
    int n = std::min((int)k.size(), cap);
    for (int i = 0; i < n; i++) { xys[3 * i] = (int)k[i].x; xys[3 * i + 1] = (int)k[i].y; xys[3 * i + 2] = (int)k[i].response; }
    return (int)k.size();
}
// threshold-free FAST strength map: score[y][x] = max(bright arc, dark arc) - 1 clamped to [0,255]; 0 outside [3,dim-3)
void oracle_fast_score_map(const uint8_t* img, int cols, int rows, int step, uint8_t* score) {
    std::memset(score, 0, (size_t)cols * rows);
    int pixel[25];
    for (int k = 0; k < 16; k++) pixel[k] = kCircle[k][0] + kCircle[k][1] * step;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
    for (int y = 3; y < rows - 3; y++)
        for (int x = 3; x < cols - 3; x++) {
            int s = corner_score16(img + (size_t)y * step + x, pixel, 0);   // threshold 0: max(0, arcs) - 1
            score[(size_t)y * cols + x] = (uint8_t)std::max(s, 0);
        }
}
float oracle_fast_atan2(float y, float x) { return fast_atan2(y, x); }
void oracle_orb_umax(int* out16) { Extractor e; for (int i = 0; i < 16; i++) out16[i] = e.umax[i]; }
// retainBest on a response list: perm_out receives the ORIGINAL positions of the survivors, in output order
int oracle_retain_best(const float* responses, int n, int n_points, int* perm_out) {
    std::vector<KeyPoint> k(n);
    for (int i = 0; i < n; i++) k[i] = KeyPoint{(float)i, 0, 0, 0, responses[i], 0, 0};
    retain_best(k, n_points);
    for (size_t i = 0; i < k.size(); i++) perm_out[i] = (int)k[i].x;
    return (int)k.size();
}
// std::nth_element permutation alone (what the GPU introselect must reproduce)
void oracle_nth_element_perm(const int* keys, int n, int nth, int* perm_out) {
    std::vector<std::pair<int, int>> v(n);
    for (int i = 0; i < n; i++) v[i] = {keys[i], i};
    std::nth_element(v.begin(), v.begin() + nth, v.end(),
                     [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first > b.first; });
    for (int i = 0; i < n; i++) perm_out[i] = v[i].second;
}

// ---- FrameExtractor's two steps around the extractor (TEST INFRASTRUCTURE like the rest of this file; parity UNPINNED: both call into
// OpenCV, which this image does not have).
// cv::cvtColor(in, gray, COLOR_BGR2GRAY) for 8-bit input (src/utils/frameextractor.cpp:2960,3046): OpenCV's RGB2Gray<uchar>, 15-bit
// fixed point — coefficients B 3735, G 19235, R 9798, rounding constant 2^14, shift 15.  cn = 3 (BGR) or 4 (BGRA, alpha ignored).
void oracle_bgr2gray(const uint8_t* src, int w, int h, size_t stride, int cn, uint8_t* dst) {
    for (int y = 0; y < h; y++) {
        const uint8_t* row = src + (size_t)y * stride;
        for (int x = 0; x < w; x++) {
            const int b = row[x * cn], g = row[x * cn + 1], r = row[x * cn + 2];
            dst[(size_t)y * w + x] = (uint8_t)((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15);
        }
    }
}
// undistortPoints(points, ImageParams) of src/basictypes/misc.cpp:269-293: cv::undistortPoints(points, out, CameraMatrix, Distorsion)
// — per point, in double: x = (u - cx) * (1 / fx); five iterations (TermCriteria(MAX_ITER, 5, 0.01)) of
//     r2 = x^2 + y^2;  icdist = (1 + ((k7 r2 + k6) r2 + k5) r2) / (1 + ((k4 r2 + k1) r2 + k0) r2);   [icdist < 0: keep the start, stop]
//     dx = 2 k2 x y + k3 (r2 + 2 x^2);  dy = k2 (r2 + 2 y^2) + 2 k3 x y;   x = (x0 - dx) icdist;  y = (y0 - dy) icdist
// the result narrowed to float — then x * fx + cx in float (misc.cpp:283-291).  cam4 = fx fy cx cy (CV_32F), dist = k1 k2 p1 p2 k3 k4 k5 k6.
void oracle_undistort_points(const float* cam4, const float* dist, int n_dist, const float* xy, int n, float* out) {
    double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n_dist && i < 8; i++) k[i] = dist[i];
    const double fx = cam4[0], fy = cam4[1], cx = cam4[2], cy = cam4[3], ifx = 1.0 / fx, ify = 1.0 / fy;
    for (int i = 0; i < n; i++) {
        double x = ((double)xy[2 * i] - cx) * ifx, y = ((double)xy[2 * i + 1] - cy) * ify;
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            if (icdist < 0) { x = x0; y = y0; break; }
            const double dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x), dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
            x = (x0 - dx) * icdist;
            y = (y0 - dy) * icdist;
        }
        const float xf = (float)x, yf = (float)y;
        volatile float px = xf * cam4[0]; volatile float py = yf * cam4[1];   // (float product, then float sum: no contraction)
        out[2 * i] = px + cam4[2];
        out[2 * i + 1] = py + cam4[3];
    }
}

}  // extern "C"
