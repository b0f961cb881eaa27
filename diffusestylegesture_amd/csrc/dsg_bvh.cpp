// dsg_bvh.cpp -- pose vector (ZEGGS, 1141-d) -> BVH, host side of libdsg_hip.so (C ABI: dsg_pose2bvh* in include/dsg.h).
//
// The step right after the sampling path.  Own C++ restatement of the reference's chain -- the formulas
// are dictated by the file format and the reference's conventions, the structure is not:
//   de-normalisation                      main/mydiffusion_zeggs/sample.py:320-326   (std clipped at 0.01, float64)
//   pose2bvh                              main/process/process_zeggs_bvh.py:219-275  (slices, Savitzky-Golay 15/2, x3 repeat)
//   xform_orthogonalize_from_xy           ubisoft-laforge-ZeroEGGS-main/ZEGGS/anim/txform.py:23-34   (float32, eps 1e-10)
//   quat.from_xform / mul / mul_vec / to_euler      .../anim/quat.py:166-206, :17-40, :111-120
//   write_bvh                             .../ZEGGS/utils_zeggs.py:47-87  (root composed into joint 0 AFTER the repeat; the
//                                         rotation array is float32, the position array float64)
//   bvh.save / save_joint                 .../anim/bvh.py:137-234         (text layout, `%f`, DFS order, End Sites)
// Savitzky-Golay (window 15, order 2, scipy mode 'interp') is one 15 x 15 projection: the hat matrix of a quadratic fit on 15
// equispaced points.  Its middle row is the interior filter, rows 0-6 / 8-14 are the edge fits of the first / last 7 frames.
// Everything per frame is independent, so a clip is processed by frame blocks on a few host threads; the text is produced by
// an exact fixed-point formatter (no printf in the inner loop) -- the whole 936-frame file takes a few milliseconds.
#include "../../include/dsg.h"
#pragma clang fp contract(off)      // the float32 steps mirror numpy / torch element-wise arithmetic: no fused multiply-adds

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

void dsg_internal_set_error(const char* msg);      // dsg_hip.cpp: the message dsg_last_error() returns

namespace {

constexpr int NJ = 75, NF = 1141, NCH = 6 + (NJ - 1) * 3;      // joints, pose features, channels per frame (228)
const int PARENTS[NJ] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 4, 9, 10, 11, 12, 13, 14, 15, 12, 17, 18, 19, 12, 21, 22, 23, 12, 25, 26,
                         27, 12, 29, 30, 31, 12, 11, 4, 35, 36, 37, 38, 39, 40, 41, 38, 43, 44, 45, 38, 47, 48, 49, 38, 51,
                         52, 53, 38, 55, 56, 57, 38, 37, 0, 61, 62, 63, 64, 63, 62, 0, 68, 69, 70, 71, 70, 69};

thread_local std::string g_bvh_err;

std::vector<std::string> bone_names() {
    std::vector<std::string> n = {"Hips", "Spine", "Spine1", "Spine2", "Spine3", "Neck", "Neck1", "Head", "HeadEnd"};
    const char* fingers[5] = {"Thumb", "Index", "Middle", "Ring", "Pinky"};
    for (const char* side : {"Right", "Left"}) {
        const std::string s(side);
        for (const char* p : {"Shoulder", "Arm", "ForeArm", "Hand"}) n.push_back(s + p);
        for (const char* f : fingers)
            for (int i = 1; i <= 4; ++i) n.push_back(s + "Hand" + f + std::to_string(i));
        n.push_back(s + "ForeArmEnd");
        n.push_back(s + "ArmEnd");
    }
    for (const char* side : {"Right", "Left"}) {
        const std::string s(side);
        for (const char* p : {"UpLeg", "Leg", "Foot", "ToeBase", "ToeBaseEnd", "LegEnd", "UpLegEnd"}) n.push_back(s + p);
    }
    return n;
}

// hat matrix of the least-squares quadratic on x = -7..7: H[i][j] = sum_{a,b} x_i^a (X^T X)^{-1}_{ab} x_j^b
struct Hat {
    double h[15][15];
    Hat() {
        // X^T X for the basis {1, x, x^2} on a symmetric grid: odd moments vanish
        double s0 = 15, s2 = 0, s4 = 0;
        for (int x = -7; x <= 7; ++x) { s2 += (double)x * x; s4 += (double)x * x * x * x; }
        const double det = s0 * s4 - s2 * s2;                  // of the {1, x^2} block
        for (int i = 0; i < 15; ++i)
            for (int j = 0; j < 15; ++j) {
                const double xi = i - 7, xj = j - 7;
                const double a = (s4 - s2 * xj * xj) / det, c = (-s2 + s0 * xj * xj) / det;      // coefficients of 1 and x^2 for e_j
                h[i][j] = a + xi * xj / s2 + c * xi * xi;
            }
    }
};
const Hat& hat() { static const Hat H; return H; }

struct Q { double w, x, y, z; };
inline Q qmul(const Q& a, const Q& b) {      // quat.mul(x, y), quat.py:17-24
    return {b.w * a.w - b.x * a.x - b.y * a.y - b.z * a.z, b.w * a.x + b.x * a.w - b.y * a.z + b.z * a.y,
            b.w * a.y + b.x * a.z + b.y * a.w - b.z * a.x, b.w * a.z - b.x * a.y + b.y * a.x + b.z * a.w};
}
inline void cross3(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

// two axes (float32, txform.py:23-34) -> quaternion (float32 arithmetic, quat.py:166-206)
inline void xy_to_quat(const float* xy, float* q) {
    const float* xa = xy;
    const float* yh = xy + 3;
    float za[3] = {xa[1] * yh[2] - xa[2] * yh[1], xa[2] * yh[0] - xa[0] * yh[2], xa[0] * yh[1] - xa[1] * yh[0]};
    float ya[3] = {za[1] * xa[2] - za[2] * xa[1], za[2] * xa[0] - za[0] * xa[2], za[0] * xa[1] - za[1] * xa[0]};
    auto nrm = [](const float* v, float* o) {
        const float n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + 1e-10f;
        o[0] = v[0] / n; o[1] = v[1] / n; o[2] = v[2] / n;
    };
    float c0[3], c1[3], c2[3];                  // COLUMNS of the rotation matrix: x, y, z axes
    nrm(xa, c0); nrm(ya, c1); nrm(za, c2);
    const float m00 = c0[0], m10 = c0[1], m20 = c0[2], m01 = c1[0], m11 = c1[1], m21 = c1[2], m02 = c2[0], m12 = c2[1], m22 = c2[2];
    const float eps = 1e-10f;
    const float t = m00 + m11 + m22;
    if (t > 0.f) {
        const float s = 0.5f / std::sqrt(std::max(t + 1.f, eps));
        q[0] = 0.25f / s; q[1] = s * (m21 - m12); q[2] = s * (m02 - m20); q[3] = s * (m10 - m01);
    } else if (m00 > m11 && m00 > m22) {
        const float s = 2.0f * std::sqrt(std::max(1.0f + m00 - m11 - m22, eps));
        q[0] = (m21 - m12) / s; q[1] = s * 0.25f; q[2] = (m01 + m10) / s; q[3] = (m02 + m20) / s;
    } else if (m11 > m22) {
        const float s = 2.0f * std::sqrt(std::max(1.0f + m11 - m00 - m22, eps));
        q[0] = (m02 - m20) / s; q[1] = (m01 + m10) / s; q[2] = s * 0.25f; q[3] = (m12 + m21) / s;
    } else {
        const float s = 2.0f * std::sqrt(std::max(1.0f + m22 - m00 - m11, eps));
        q[0] = (m10 - m01) / s; q[1] = (m02 + m20) / s; q[2] = (m12 + m21) / s; q[3] = s * 0.25f;
    }
}
// float32 quaternion -> Euler zyx in degrees (quat.py:111-120 on a float32 array, np.degrees)
inline void quat_to_euler_deg(const float* q, double* e) {
    const double x0 = q[0], x1 = q[1], x2 = q[2], x3 = q[3];
    const float r2d = (float)(180.0 / M_PI);
    const float a = (float)std::atan2(2.0 * (x0 * x3 + x1 * x2), 1.0 - 2.0 * (x2 * x2 + x3 * x3));
    const float b = (float)std::asin(std::min(1.0, std::max(-1.0, 2.0 * (x0 * x2 - x3 * x1))));
    const float c = (float)std::atan2(2.0 * (x0 * x1 + x2 * x3), 1.0 - 2.0 * (x1 * x1 + x2 * x2));
    e[0] = (double)(a * r2d); e[1] = (double)(b * r2d); e[2] = (double)(c * r2d);
}

// de-normalise + smooth one frame: out[NF] (float64)
struct Clip {
    const void* poses; int dtype, frames; const double* mean; const double* std; int smoothing;
    double at(int f, int c) const {
        const double v = dtype == 0 ? (double)((const float*)poses)[(size_t)f * NF + c] : ((const double*)poses)[(size_t)f * NF + c];
        return mean ? v * std::max(std[c], 0.01) + mean[c] : v;
    }
};
void frame_row(const Clip& c, int f, double* out) {
    if (!c.smoothing) { for (int k = 0; k < NF; ++k) out[k] = c.at(f, k); return; }
    const Hat& H = hat();
    const int F = c.frames;
    int base, row;                                       // 15-frame support and the hat-matrix row that evaluates frame f on it
    if (f < 7) { base = 0; row = f; }
    else if (f >= F - 7) { base = F - 15; row = f - (F - 15); }
    else { base = f - 7; row = 7; }
    for (int k = 0; k < NF; ++k) {
        double acc = 0.0;
        for (int j = 0; j < 15; ++j) acc += H.h[row][j] * c.at(base + j, k);
        out[k] = acc;
    }
}

// one pose frame -> offsets-style data: positions [NJ][3] (float64, root composed), euler degrees [NJ][3]
void frame_channels(const double* p, double* pos, double* rot) {
    const double* root_pos = p;
    const Q root_rot = {p[3], p[4], p[5], p[6]};
    const double* lpos = p + 13;
    const double* ltxy = p + 13 + NJ * 3;
    for (int j = 0; j < NJ; ++j) {
        float xy[6], q[4];
        for (int k = 0; k < 6; ++k) xy[k] = (float)ltxy[j * 6 + k];
        xy_to_quat(xy, q);
        if (j == 0) {      // utils_zeggs.py:73-74
            const double qv[3] = {root_rot.x, root_rot.y, root_rot.z};
            double t[3], u[3];
            cross3(qv, lpos, t);
            for (double& v : t) v *= 2.0;
            cross3(qv, t, u);
            for (int k = 0; k < 3; ++k) pos[k] = (lpos[k] + root_rot.w * t[k] + u[k]) + root_pos[k];
            const Q r = qmul(root_rot, Q{q[0], q[1], q[2], q[3]});
            q[0] = (float)r.w; q[1] = (float)r.x; q[2] = (float)r.y; q[3] = (float)r.z;      // stored into the float32 rotation array
        } else {
            for (int k = 0; k < 3; ++k) pos[j * 3 + k] = lpos[j * 3 + k];
        }
        quat_to_euler_deg(q, rot + j * 3);
    }
}

// "%f" (6 decimals, round-to-nearest of the exact binary value) without printf: v * 1e6 is formed exactly as p + e with one
// fma, so the rounding decision is exact.  Falls back to snprintf outside the fast range.
inline char* fmt_f(char* o, double v) {
    if (!(std::fabs(v) < 1e9)) return o + std::snprintf(o, 400, "%f", v);
    const bool neg = std::signbit(v);
    const double a = std::fabs(v);
    const double p = a * 1e6, e = std::fma(a, 1e6, -p);
    double n = std::floor(p);
    double fr = (p - n) + e;
    if (fr < 0.0) { n -= 1.0; fr += 1.0; }
    if (fr > 0.5 || (fr == 0.5 && std::fmod(n, 2.0) == 1.0)) n += 1.0;
    unsigned long long u = (unsigned long long)n;
    if (neg) *o++ = '-';                                   // Python prints -0.000000 for negative values that round to zero too
    char tmp[24];
    int len = 0;
    do { tmp[len++] = (char)('0' + u % 10); u /= 10; } while (u);
    while (len < 7) tmp[len++] = '0';
    for (int i = len - 1; i >= 6; --i) *o++ = tmp[i];
    *o++ = '.';
    for (int i = 5; i >= 0; --i) *o++ = tmp[i];
    return o;
}

void dfs(int i, const std::vector<std::vector<int>>& kids, std::vector<int>& seq) {
    seq.push_back(i);
    for (int k : kids[i]) dfs(k, kids, seq);
}
// children lists and the depth-first joint order of the file; built once (function-local static: initialisation is
// thread-safe -- the batch writer enters here from many threads at once)
struct Skeleton {
    std::vector<std::vector<int>> kids;
    std::vector<int> seq;
    Skeleton() {
        kids.assign(NJ, {});
        for (int j = 1; j < NJ; ++j) kids[PARENTS[j]].push_back(j);
        dfs(0, kids, seq);
    }
};
const Skeleton& skeleton() { static const Skeleton s; return s; }
const std::vector<int>& joint_sequence(std::vector<std::vector<int>>* kids_out = nullptr) {
    const Skeleton& s = skeleton();
    if (kids_out) *kids_out = s.kids;
    return s.seq;
}

// offsets [NJ*3], motion [3*frames][NCH] in file order (root: 3 positions + 3 rotations, then 3 rotations per joint in DFS order)
int channels(const Clip& c, double* offsets, double* motion) {
    if (c.frames <= 0) { g_bvh_err = "pose2bvh: no frames"; return DSG_E_INVALID; }
    if (c.smoothing && c.frames < 15) { g_bvh_err = "pose2bvh: Savitzky-Golay window (15) exceeds the clip length"; return DSG_E_INVALID; }
    const std::vector<int>& seq = joint_sequence();
    const int F = c.frames;
    const int nthr = (int)std::max(1u, std::min(8u, std::min(std::thread::hardware_concurrency(), (unsigned)(F / 64 + 1))));
    auto work = [&](int f0, int f1) {
        std::vector<double> row(NF), pos(NJ * 3), rot(NJ * 3);
        for (int f = f0; f < f1; ++f) {
            frame_row(c, f, row.data());
            frame_channels(row.data(), pos.data(), rot.data());
            if (f == 0 && offsets) std::memcpy(offsets, pos.data(), sizeof(double) * NJ * 3);
            double* m = motion + (size_t)3 * f * NCH;
            int k = 0;
            for (int j : seq) {
                if (j == 0) { m[k++] = pos[0]; m[k++] = pos[1]; m[k++] = pos[2]; }
                m[k++] = rot[j * 3]; m[k++] = rot[j * 3 + 1]; m[k++] = rot[j * 3 + 2];
            }
            std::memcpy(m + NCH, m, sizeof(double) * NCH);          // 20 fps -> 60 fps: every frame three times
            std::memcpy(m + 2 * NCH, m, sizeof(double) * NCH);
        }
    };
    if (nthr == 1) work(0, F);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthr; ++t) th.emplace_back(work, (int)((long long)F * t / nthr), (int)((long long)F * (t + 1) / nthr));
        for (auto& t : th) t.join();
    }
    return 0;
}

void joint_text(int i, std::string t, const std::vector<std::vector<int>>& kids, const std::vector<std::string>& names,
                const double* offsets, std::string& out) {
    char buf[512];
    out += t + "JOINT " + names[i] + "\n" + t + "{\n";
    t += '\t';
    std::snprintf(buf, sizeof buf, "%sOFFSET %f %f %f\n", t.c_str(), offsets[i * 3], offsets[i * 3 + 1], offsets[i * 3 + 2]);
    out += buf;
    out += t + "CHANNELS 3 Zrotation Yrotation Xrotation\n";
    for (int k : kids[i]) joint_text(k, t, kids, names, offsets, out);
    if (kids[i].empty()) {
        out += t + "End Site\n" + t + "{\n";
        std::snprintf(buf, sizeof buf, "%s\tOFFSET %f %f %f\n", t.c_str(), 0.0, 0.0, 0.0);
        out += buf;
        out += t + "}\n";
    }
    t.pop_back();
    out += t + "}\n";
}

int write_file(const Clip& c, const char* path) {
    const int F3 = 3 * c.frames;
    std::vector<double> offsets(NJ * 3), motion((size_t)F3 * NCH);
    const int rc = channels(c, offsets.data(), motion.data());
    if (rc) return rc;
    std::vector<std::vector<int>> kids;
    joint_sequence(&kids);
    const std::vector<std::string> names = bone_names();
    std::string text = "HIERARCHY\nROOT " + names[0] + "\n{\n";
    char buf[512];
    std::snprintf(buf, sizeof buf, "\tOFFSET %f %f %f\n", offsets[0], offsets[1], offsets[2]);
    text += buf;
    text += "\tCHANNELS 6 Xposition Yposition Zposition Zrotation Yrotation Xrotation \n";
    for (int k : kids[0]) joint_text(k, "\t", kids, names, offsets.data(), text);
    text += "}\nMOTION\n";
    std::snprintf(buf, sizeof buf, "Frames: %i\nFrame Time: %f\n", F3, 1.0 / 60.0);
    text += buf;
    const size_t head = text.size();
    // every distinct frame is formatted once and written three times
    std::vector<char> line((size_t)NCH * 24 + 8);
    text.reserve(head + (size_t)F3 * NCH * 12);
    for (int f = 0; f < c.frames; ++f) {
        const double* m = motion.data() + (size_t)3 * f * NCH;
        char* o = line.data();
        for (int k = 0; k < NCH; ++k) { o = fmt_f(o, m[k]); *o++ = ' '; }
        *o++ = '\n';
        for (int r = 0; r < 3; ++r) text.append(line.data(), (size_t)(o - line.data()));
    }
    FILE* fp = std::fopen(path, "wb");
    if (!fp) { g_bvh_err = std::string("pose2bvh: cannot open ") + path; return DSG_E_RUNTIME; }
    const size_t w = std::fwrite(text.data(), 1, text.size(), fp);
    if (std::fclose(fp) != 0 || w != text.size()) { g_bvh_err = std::string("pose2bvh: short write to ") + path; return DSG_E_RUNTIME; }
    return 0;
}

}  // namespace

static int done(int rc) { if (rc) dsg_internal_set_error(g_bvh_err.c_str()); return rc; }

extern "C" int dsg_pose2bvh_channels(const void* poses, int dtype, int frames, const double* mean, const double* std_, int smoothing,
                                     double* offsets, double* motion) {
    if (!poses || !motion || (dtype != 0 && dtype != 1) || ((mean == nullptr) != (std_ == nullptr))) { g_bvh_err = "pose2bvh: bad argument"; return done(DSG_E_INVALID); }
    const Clip c = {poses, dtype, frames, mean, std_, smoothing};
    return done(channels(c, offsets, motion));
}

extern "C" int dsg_pose2bvh(const void* poses, int dtype, int frames, const double* mean, const double* std_, int smoothing, const char* outpath) {
    if (!poses || !outpath || (dtype != 0 && dtype != 1) || ((mean == nullptr) != (std_ == nullptr))) { g_bvh_err = "pose2bvh: bad argument"; return done(DSG_E_INVALID); }
    const Clip c = {poses, dtype, frames, mean, std_, smoothing};
    return done(write_file(c, outpath));
}

// n clips (same length), one file each, clips spread over host threads: the serial tail of a many-clip job
extern "C" int dsg_pose2bvh_batch(const void* poses, int dtype, int n_clips, int frames, const double* mean, const double* std_, int smoothing,
                                  const char* const* outpaths) {
    if (!poses || !outpaths || n_clips <= 0 || (dtype != 0 && dtype != 1) || ((mean == nullptr) != (std_ == nullptr))) { g_bvh_err = "pose2bvh: bad argument"; return done(DSG_E_INVALID); }
    const size_t stride = (size_t)frames * NF * (dtype == 0 ? 4 : 8);
    std::vector<int> rcs(n_clips, 0);
    std::vector<std::string> errs(n_clips);
    const int nthr = (int)std::max(1u, std::min((unsigned)n_clips, std::min(32u, std::thread::hardware_concurrency())));
    std::vector<std::thread> th;
    for (int t = 0; t < nthr; ++t)
        th.emplace_back([&, t]() {
            for (int i = t; i < n_clips; i += nthr) {
                const Clip c = {(const char*)poses + stride * i, dtype, frames, mean, std_, smoothing};
                rcs[i] = write_file(c, outpaths[i]);
                if (rcs[i]) errs[i] = g_bvh_err;
            }
        });
    for (auto& t : th) t.join();
    for (int i = 0; i < n_clips; ++i)
        if (rcs[i]) { g_bvh_err = errs[i]; return done(rcs[i]); }
    return 0;
}
