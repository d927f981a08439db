// FrameMatcher_Flann post-filter (host code): what UcoSLAM does with the kNN rows the index returns.
//
// Reference: src/utils/framematcher.cpp:228-319 (matchEpipolar body after the search), :67-108 (computeThreeMaxima),
//            src/basictypes/misc.cpp:105-107 (remove_unused_matches), :153-185 (filter_ambiguous_train), :117-150
//            (filter_ambiguous_query), src/basictypes/misc.h:72-81 (epipolarLineSqDist).
// This is sequential policy over <= nq*nn candidates (microseconds on a host core); it stays on the host exactly like the
// reference, consuming the (bit-exact, heap-ordered) rows of uh_knn_search.  The result depends on the column ORDER of the
// unsorted rows, which is why uh_knn_search reproduces the reference heap layout.
#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

#include "common.hpp"

namespace {

inline float epipolar_sq_dist(const float* kp1, const float* kp2, const float* F) {   // misc.h:72-81, F row-major 3x3
    const float a = kp1[0] * F[0] + kp1[1] * F[3] + F[6];
    const float b = kp1[0] * F[1] + kp1[1] * F[4] + F[7];
    const float den = a * a + b * b;
    if (den == 0) return std::numeric_limits<float>::max();
    const float c = kp1[0] * F[2] + kp1[1] * F[5] + F[8];
    const float num = a * kp2[0] + b * kp2[1] + c;
    return num * num / den;
}

void remove_unused(std::vector<uh_dmatch>& m) {
    m.erase(std::remove_if(m.begin(), m.end(), [](const uh_dmatch& x) { return x.trainIdx == -1 || x.queryIdx == -1; }), m.end());
}

void filter_ambiguous(std::vector<uh_dmatch>& matches, bool by_train) {
    if (matches.empty()) return;
    int maxT = -1;
    for (const auto& m : matches) maxT = std::max(maxT, by_train ? m.trainIdx : m.queryIdx);
    std::vector<int> used(maxT + 1, -1);
    int idx = 0;
    bool needRemove = false;
    for (auto& match : matches) {
        int& key = by_train ? match.trainIdx : match.queryIdx;
        if (used[key] == -1) used[key] = idx;
        else {
            uh_dmatch& other = matches[used[key]];
            if (other.distance > match.distance) {
                (by_train ? other.trainIdx : other.queryIdx) = -1;   // annulate the other match
                used[key] = idx;
                needRemove = true;
            } else {
                key = -1;                                            // annulate this match
                needRemove = true;
            }
        }
        idx++;
    }
    if (needRemove) remove_unused(matches);
}

}  // namespace

extern "C" {

int uh_filter_ambiguous(uh_dmatch* matches, int n, int by_train) {
    UH_REQUIRE(n >= 0 && (n == 0 || matches), "uh_filter_ambiguous: bad arguments");
    std::vector<uh_dmatch> v(matches, matches + n);
    filter_ambiguous(v, by_train != 0);
    std::copy(v.begin(), v.end(), matches);
    return (int)v.size();
}

int uh_match_filter(const uh_match_filter_args* a, uh_dmatch* out, int cap) {
    UH_REQUIRE(a && out, "uh_match_filter: NULL argument");
    UH_REQUIRE(a->nq >= 0 && a->nn >= 1 && a->indices && a->distances, "uh_match_filter: bad kNN rows");
    UH_REQUIRE(a->q_octave && a->q_angle && a->t_octave && a->t_angle, "uh_match_filter: keypoint arrays missing");
    if (a->F12) UH_REQUIRE(a->q_pt && a->t_pt && a->scale_factors, "uh_match_filter: epipolar gate needs points and scale factors");
    std::vector<uh_dmatch> matches;
    for (int i = 0; i < a->nq; i++) {
        float bestDist = a->min_desc_dist, bestDist2 = std::numeric_limits<float>::max();
        long bestQuery = -1, bestTrain = -1;
        int octaveBest2 = -1;
        const int queryIndex = a->map_idx_query ? (int)a->map_idx_query[i] : i;
        for (int j = 0; j < a->nn; j++) {
            const float dist = (float)a->distances[(size_t)i * a->nn + j];   // distances.convertTo(CV_32F)
            if (dist > a->min_desc_dist) continue;
            if (dist < bestDist2) {
                const int ti = a->indices[(size_t)i * a->nn + j];
                // xflann leaves unfilled slots at index -1 (distance 0); the reference would index map_idx_trainkp[-1] here.
                if (ti < 0) continue;
                const int trainIdx = a->map_idx_train ? (int)a->map_idx_train[ti] : ti;
                if (std::abs(a->t_octave[trainIdx] - a->q_octave[queryIndex]) > a->max_octave_diff) continue;
                if (a->F12) {
                    const float s = a->scale_factors[a->q_octave[queryIndex]];
                    if (epipolar_sq_dist(a->t_pt + 2 * (size_t)trainIdx, a->q_pt + 2 * (size_t)queryIndex, a->F12) >= 3.84 * (double)(s * s)) continue;
                }
                if (dist < bestDist) { bestDist = dist; bestQuery = queryIndex; bestTrain = trainIdx; }
                else { bestDist2 = dist; octaveBest2 = a->t_octave[trainIdx]; }
            }
        }
        if (bestQuery != -1) {
            if (!(octaveBest2 == a->q_octave[bestQuery] && bestDist > bestDist2 * a->nn_match_ratio)) {
                uh_dmatch m;
                m.queryIdx = (int)bestQuery; m.trainIdx = (int)bestTrain; m.imgIdx = -1; m.distance = bestDist;
                matches.push_back(m);
            }
        }
    }
    filter_ambiguous(matches, true);
    if (a->check_orientation) {
        std::vector<std::vector<int>> rotHist(30);
        const float factor = 1.0f / float(rotHist.size());
        for (size_t midx = 0; midx < matches.size(); midx++) {
            float rot = a->t_angle[matches[midx].trainIdx] - a->q_angle[matches[midx].queryIdx];
            if (rot < 0.0) rot += 360.0f;
            size_t bin = (size_t)std::round(rot * factor);
            if (bin == rotHist.size()) bin = 0;
            rotHist[bin].push_back((int)midx);
        }
        int ind1 = -1, ind2 = -1, ind3 = -1;
        {   // computeThreeMaxima
            int max1 = 0, max2 = 0, max3 = 0;
            for (size_t i = 0; i < rotHist.size(); i++) {
                const int s = (int)rotHist[i].size();
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = (int)i; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = (int)i; }
                else if (s > max3) { max3 = s; ind3 = (int)i; }
            }
            if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if (max3 < 0.1f * (float)max1) ind3 = -1;
        }
        for (int i = 0; i < (int)rotHist.size(); i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int midx : rotHist[i]) matches[midx].queryIdx = matches[midx].trainIdx = -1;
        }
        remove_unused(matches);
    }
    if ((int)matches.size() > cap) { uh::set_error("uh_match_filter: %zu matches but capacity %d", matches.size(), cap); return UH_ECAPACITY; }
    std::copy(matches.begin(), matches.end(), out);
    return (int)matches.size();
}

}  // extern "C"
