// mrca_policy.hip -- the lidar front end of the actor-critic (model/net.py:19-25,37-49: two Conv1d + ReLU per tower)
// as ONE fused gfx950 kernel: the rollout's inference and the forward half of the PPO update (its backward pass is
// mrca_policy_bwd.hip).
//
//   feat[t][n][c*128 + l] = relu(b2[t][c] + sum_{ci<32,k<3} w2[t][c][ci][k] * h1[t][ci][2l+k-1])          l < 128
//   h1[t][c][l]           = relu(b1[t][c] + sum_{ci<3, k<5} w1[t][c][ci][k] * obs[n][ci][2l+k-1])         l < 255
//   (Conv1d(3,32,k5,s2,p1) -> ReLU -> Conv1d(32,32,k3,s2,p1) -> ReLU, towers t = actor, critic; zero padding.)
//
// Why a kernel: the stock path runs these two tiny-channel convolutions through MIOpen at ~10 TFLOP/s -- two thirds
// of a rollout tick at 4096 robots -- and round-trips the 32 x 255 intermediate (268 MB per tick for both towers)
// through HBM.  Here one wavefront owns one (robot, tower): the scan is staged de-interleaved in LDS (stride-2
// convolutions become unit-stride reads), both convolutions are fp32 MFMA GEMMs (v_mfma_f32_32x32x2_f32: exact f32,
// a k-ordered fmaf chain) on implicit im2col operands read straight from LDS, the intermediate never leaves the CU,
// and the weights of the tower live in registers for the wave's whole life (persistent waves walk the robots).
//   conv1:  C[32 ch][256 pos] = W1[32][16] x X1[16][256]     (K = 3*5 = 15 + the bias as a 16th k)   64 MFMAs
//   conv2:  C[32 ch][128 pos] = W2[32][96] x X2[96][128]     (K = 32*3)                              192 MFMAs
// 256 MFMAs x 64 cycles per (robot, tower): 55 us at 4096 robots if every SIMD issued back to back.
//
// What keeps the matrix pipe fed (round 3; the round-2 kernel reached 50 % of the fp32 MFMA peak -- its ISA waited on
// every LDS read right before the MFMA that used it and staged the scan with six serialised HBM round trips):
//   * the next robot's scan is requested from HBM as soon as the current one is staged;
//   * every LDS operand is requested a chunk (8 MFMAs) ahead of its use, the order pinned with sched_barrier;
//   * the K index is enumerated so that the two k of an MFMA step differ by a constant address offset
//     (mrca_policy_layout.h): all operand addresses are "one of five lane-constant bases + immediate";
//   * conv2 runs as two tile pairs; the first pair's output leaves through LDS as 16-byte rows (one float4 store per lane
//     covers 4 rows x 256 contiguous bytes instead of a dword store per register) while the second pair's MFMAs run.
//
// fp32 in, fp32 accumulate: the result differs from the PyTorch layers only by summation order (tested to 1e-5).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mrca_env.h"
#include "mrca_hostutil.h"
#include "mrca_policy_layout.h"

namespace mrca_policy {

using namespace mrca_pfwd;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;     // (HIP's float4 is a struct around a union: arrays of it stay in scratch)

#define MRCA_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MRCA_PIN() __builtin_amdgcn_sched_barrier(0)

__device__ inline f32x16 splat16(const float (&v)[16]) {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = v[r];
    return z;
}

// ReLU as ONE integer instruction: the bit pattern of a negative float is a negative integer (-0.0 included), that of a
// positive float a positive one, so relu(x) = as_float(max(as_int(x), 0)).  (x > 0 ? x : 0 compiles to two v_max_f32 under
// IEEE rules; a negative NaN becomes 0 here -- the layers never produce one from finite inputs.)
__device__ __forceinline__ float relu(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

__device__ inline f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
    return z;
}

#if defined(MRCA_PROFILING)
// profiling build only: s_memtime ticks a wave spends in each phase of a robot (summed over its robots) + robot count;
// reading the clock drains the LDS queue, so the stamped kernel runs a few per cent slower than the product
constexpr int kFwdStamps = 8, kFwdStampWaves = 1024;
__device__ unsigned long long g_fwd_stamps[kFwdStamps][kFwdStampWaves];
#define MRCA_FSTAMP(k)                                              \
    {                                                               \
        MRCA_PIN();                                                 \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        fst[k] += t_ - fprev;                                       \
        fprev = t_;                                                 \
        MRCA_PIN();                                                 \
    }
#else
#define MRCA_FSTAMP(k)
#endif

// Where robot n's three frames are, as rows of 512 floats behind `obs`, oldest first:
//   rows != NULL: the caller's table, rows[3 n + f] (the rollout buffer's one-frame-per-tick store read in place, mrca/ppo.py)
//   head != NULL: `obs` is [n][3][512] used as a ring: frame f sits in slot (head[n] + 1 + f) mod 3
//   neither:      `obs` is [n][3][512] in deque order
// Fetched one robot ahead of the scan it describes (a dependent load in front of six float4 loads otherwise).
struct FrameRows3 {
    int r0, r1, r2;
};
__device__ __forceinline__ FrameRows3 frame_rows(const uint8_t* __restrict__ head, const int32_t* __restrict__ rows, int n) {
    if (rows) return FrameRows3{rows[3 * n], rows[3 * n + 1], rows[3 * n + 2]};
    const int hd = head ? head[n] : 2;                                  // hd = 2: the identity (deque order)
    const int s0 = hd == 2 ? 0 : hd + 1, s1 = s0 == 2 ? 0 : s0 + 1;
    return FrameRows3{3 * n + s0, 3 * n + s1, 3 * n + hd};
}
// the six float4 a lane loads of a robot's scan: float4 q covers logical frame q / 2
__device__ __forceinline__ void request_scan(float4 (&sx)[6], const float* __restrict__ obs, FrameRows3 fr, int lane) {
    const float4* src = reinterpret_cast<const float4*>(obs);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int row = (q >> 1) == 0 ? fr.r0 : ((q >> 1) == 1 ? fr.r1 : fr.r2);
        sx[q] = src[(size_t)row * (kBeams / 4) + (q & 1) * 64 + lane];
    }
}

// x / 6 - 0.5 (stage_world1.py:140) exactly as the env's own views compute it (mrca_device.h:norm_obs -- restated here
// because this translation unit does not see the env's headers; tests/test_gpu_policy_ops.py holds the two bit-identical)
__device__ __forceinline__ float norm_scan(float x) {
    const float inv6 = 1.0f / 6.0f;
    const float q = x * inv6;
    const float r = __builtin_fmaf(-q, 6.0f, x);
    return __builtin_fmaf(r, inv6, q) - 0.5f;
}

// RAW: `obs` holds raw ranges (the env's ring of scans); the observation is formed while the scan is staged
template <bool RAW>
__global__ __launch_bounds__(64 * kWavesPerBlock) void lidar_features_kernel(
    const float* __restrict__ obs, const uint8_t* __restrict__ head, const int32_t* __restrict__ rows, int n_robots,
    const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ feat) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float* lds = lds_all + wave * kWaveFloats;
    const int gwave = blockIdx.x * kWavesPerBlock + wave;
    const int nwaves = gridDim.x * kWavesPerBlock;
    const int tower = gwave & 1;                 // waves come in (actor, critic) pairs on the same robots
    const int col = lane & 31, hl = lane >> 5;

    // --- the tower's weights as MFMA A fragments A[i = out channel = col][k], in the K orders of mrca_policy_layout.h.
    // Fetched COALESCED (float4 per lane, 12 + 2 instructions) into this wave's still unused H1 area, rows padded to odd
    // pitches, and picked up from there: as 56 loads with a lane stride of 96 / 15 floats every instruction touched 64 cache
    // lines -- ~7 us of a launch's ~13 us outside the robot loop, four waves behind one L1 (profiles/r03/r03_n_fwd_phases.txt).
    float a1[8], a2[48];
    {
        constexpr int kW2L = kH1E, kW1L = kH1E + 32 * 97;       // [32][97], [32][17]
        static_assert(kW1L + 32 * 17 <= kWaveFloats, "the weight staging fits the H1 area");
        const float4* w2v = reinterpret_cast<const float4*>(w2 + tower * 3072);
        const float4* w1v = reinterpret_cast<const float4*>(w1 + tower * 480);
        float4 t2[12], t1[2];
#pragma unroll
        for (int q = 0; q < 12; ++q) t2[q] = w2v[q * 64 + lane];
        t1[0] = w1v[lane];
        t1[1] = w1v[lane < 56 ? 64 + lane : 64];
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const int f = q * 64 + lane;                          // float4 index: row f / 24, columns 4 (f % 24) ...
            float* d = lds + kW2L + (f / 24) * 97 + 4 * (f % 24);
            d[0] = t2[q].x;
            d[1] = t2[q].y;
            d[2] = t2[q].z;
            d[3] = t2[q].w;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (q == 0 || lane < 56) {
                const float v[4] = {t1[q].x, t1[q].y, t1[q].z, t1[q].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = 4 * (q * 64 + lane) + j;        // element: row e / 15, column e % 15
                    lds[kW1L + (e / 15) * 17 + e % 15] = v[j];
                }
            }
        }
        const float bias1 = b1[tower * 32 + col];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int kk = conv1_kk(s, hl);
            a1[s] = kk < 15 ? lds[kW1L + col * 17 + kk] : bias1;
        }
#pragma unroll
        for (int s = 0; s < 48; ++s) a2[s] = lds[kW2L + col * 97 + conv2_ci(s, hl) * 3 + conv2_tap(s, hl)];
    }
    float bias2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias2[r] = b2[tower * 32 + rowmap(r, hl)];

    // constant parts of this wave's LDS image: the paddings x[ci][-1], h1[c][-1], h1[c][255] and the row tails the
    // last conv1 tile reads for the non-existent position 255
    if (lane < 3) {
        lds[kXO + lane * kXPitch] = 0.0f;
#pragma unroll
        for (int k = 256; k < kXPitch; ++k) lds[kXE + lane * kXPitch + k] = 0.0f;
#pragma unroll
        for (int k = 257; k < kXPitch; ++k) lds[kXO + lane * kXPitch + k] = 0.0f;
    }
    if (lane < kCh) {
        lds[kH1O + lane * kHPitch] = 0.0f;
        lds[kH1O + lane * kHPitch + 128] = 0.0f;
    }

    // lane-constant bases; every access below is <base>[<compile-time offset>]
    const float* x1 = lds + col + hl;                    // conv1 operand families (mrca_policy_layout.h)
    const float* x2 = lds + col + hl * kXPitch;
    const float* x3 = lds + col;
    float* hst = lds + ((col & 1) ? kH1O + (col + 1) / 2 : kH1E + col / 2) + 4 * hl * kHPitch;   // h1_store_off + row
    const float* ha = lds + col + hl;                    // conv2 operand families
    const float* hb = lds + col + hl * kHPitch;
    float* oe = lds + kH1E + 4 * hl * kHPitch + col;     // epilogue: accumulators in, [channel][position]
    const float* orow = lds + kH1E + (lane >> 4) * kHPitch + 4 * (lane & 15);   // ... rows out: float4 q -> channel 4q + lane/16
    const size_t gofs = (size_t)(lane >> 4) * kL2 + 4 * (lane & 15);

    const int stride = nwaves >> 1;
    int n = gwave >> 1;
    if (n >= n_robots) return;       // wave-uniform; the kernel has no barrier

    // ---- software pipeline (round 3, v3).  One wave per SIMD (the LDS image decides that), so everything that is not an
    // MFMA has to ISSUE IN THE SHADOW of one: an MFMA occupies the matrix pipe for 64 cycles and the wave may issue ~12
    // other instructions meanwhile.  Per robot, in program order (LDS operations of one wave complete in order, so
    // program order is the only hazard rule):
    //   conv1 pair 0   | + the PREVIOUS robot's conv2 pair 1 leaves: accumulators -> H1E[c][64..127]
    //   conv1 pair 1   | + pair 0's ReLU + stores to H1; those rows back as float4 (registers)
    //   conv1 pair 2   | + pair 1's stores; the float4 rows out to HBM
    //   conv1 pair 3   | + pair 2's stores (these overwrite H1E[c][64..]: the rows were read two pairs ago)
    //   conv2 pair 0   | + pair 3's stores (conv2 pair 0 reads positions <= 127 = conv1 pairs 0, 1 only);
    //                  |   the NEXT robot's scan -> XE / XO (conv1 is through with them), the one after requested from HBM
    //   conv2 pair 1   | + pair 0's output through H1E[c][0..63] and out; the next robot's first conv1 operands
    // Every LDS operand is requested a chunk ahead, across the phase boundaries too.
    float4 sx[6];
    request_scan(sx, obs, frame_rows(head, rows, n), lane);
    FrameRows3 fr_next = frame_rows(head, rows, n + stride < n_robots ? n + stride : n);

#define MRCA_STAGE_SCAN(q)                                                                               \
    {                                                                                                    \
        const int idx = (q) * 64 + lane; /* float4 index: ci = idx / 128, m = idx % 128 -> x[ci][4m .. 4m+3] */ \
        float4 v = sx[q];                                                                                \
        if (RAW) v = make_float4(norm_scan(fabsf(v.x)), norm_scan(fabsf(v.y)), norm_scan(fabsf(v.z)), norm_scan(fabsf(v.w))); /* |x|: rows stored under ABI 4-5 carried a flag in the sign bit */ \
        const int ci = idx >> 7, m = idx & 127;                                                          \
        float* xe = lds + kXE + ci * kXPitch + 2 * m;                                                    \
        float* xo = lds + kXO + ci * kXPitch + 2 * m + 1;                                                \
        xe[0] = v.x;                                                                                     \
        xo[0] = v.y;                                                                                     \
        xe[1] = v.z;                                                                                     \
        xo[1] = v.w;                                                                                     \
    }
#define MRCA_REQUEST_NEXT()                                                                              \
    if (n + stride < n_robots) {                                                                         \
        request_scan(sx, obs, fr_next, lane);                                                            \
        if (n + 2 * stride < n_robots) fr_next = frame_rows(head, rows, n + 2 * stride);                 \
    }
#define MRCA_REQUEST_NEXT_AFTER() /* inside the loop: robot n + stride has just been staged */             \
    if (n + 2 * stride < n_robots) {                                                                     \
        request_scan(sx, obs, fr_next, lane);                                                            \
        if (n + 3 * stride < n_robots) fr_next = frame_rows(head, rows, n + 3 * stride);                 \
    }
#define MRCA_CONV1_LOAD(buf, T)                                                                          \
    _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) {                                                   \
        const float* base = conv1_family(s_) == 1 ? x1 : (conv1_family(s_) == 2 ? x2 : x3);              \
        ba[buf][s_] = base[conv1_step_off(s_) + 32 * (T)];                                               \
        bb[buf][s_] = base[conv1_step_off(s_) + 32 * (T) + 32];                                          \
    }                                                                                                    \
    ba[buf][7] = hl ? 1.0f : ba[buf][7];                                                                 \
    bb[buf][7] = hl ? 1.0f : bb[buf][7];
// ReLU + store of register r of conv1 pair tpp.  (Position 255 does not exist and its slot H1O[c][128] is conv2's right
// padding: the last tile stores there like everywhere else -- no divergent branch -- and MRCA_H1_PAD puts the zero back
// before conv2's pair 1 reads it.)
#define MRCA_CONV1_OUT(tpp, r)                                                                           \
    {                                                                                                    \
        hst[32 * (tpp) + rowmap(r, 0) * kHPitch] = relu(c1a[(tpp) & 1][r]);                              \
        hst[32 * (tpp) + 16 + rowmap(r, 0) * kHPitch] = relu(c1b[(tpp) & 1][r]);                         \
    }
#define MRCA_H1_PAD() lds[kH1O + col * kHPitch + 128] = 0.0f;
// chunk g = 12 P + ch of conv2's 24 chunks of four K steps: operands of step k of the tile pair P
#define MRCA_CONV2_LOAD_K(buf, g, k)                                                                     \
    {                                                                                                    \
        const int s_ = 4 * ((g) % 12) + (k);                                                             \
        const float* base = s_ < 32 ? ha : hb;                                                           \
        b0[buf][k] = base[conv2_step_off(s_) + 64 * ((g) / 12)];                                         \
        b1v[buf][k] = base[conv2_step_off(s_) + 64 * ((g) / 12) + 32];                                   \
    }
#define MRCA_CONV2_LOAD(buf, g) _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) MRCA_CONV2_LOAD_K(buf, g, k_)
// accumulators of a conv2 tile pair -> H1E[c][64 P + ...], registers [r0, r0 + cnt)
#define MRCA_CONV2_OUT(A0, A1, P, r0, cnt)                                                               \
    _Pragma("unroll") for (int r = (r0); r < (r0) + (cnt); ++r) {                                        \
        oe[rowmap(r, 0) * kHPitch + 64 * (P)] = relu(A0[r]);                                             \
        oe[rowmap(r, 0) * kHPitch + 64 * (P) + 32] = relu(A1[r]);                                        \
    }

#pragma unroll
    for (int q = 0; q < 6; ++q) MRCA_STAGE_SCAN(q)
    MRCA_PIN();
    MRCA_REQUEST_NEXT()
    MRCA_PIN();
    float ba[2][8], bb[2][8];
    MRCA_CONV1_LOAD(0, 0)
    MRCA_PIN();

#if defined(MRCA_PROFILING)
    unsigned long long fst[kFwdStamps] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long fprev = __builtin_amdgcn_s_memtime();
    const unsigned long long freal0 = __builtin_amdgcn_s_memrealtime();      // the constant 100 MHz counter
#endif
    f32x16 p1a = zero16(), p1b = zero16();     // conv2 pair 1 of the previous robot, still to leave
    float* out_prev = feat;
    bool have_prev = false;
    for (; n < n_robots; n += stride) {
        float* out = feat + ((size_t)tower * n_robots + n) * (kCh * kL2) + gofs;
        f32x16 c1a[2], c1b[2];
        f32x4 row[8];
        float b0[2][4], b1v[2][4];

        // --- conv1: 8 position tiles of 32 in pairs
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            const int cur = tp & 1, nxt = cur ^ 1;
            if (tp < 3) {
                MRCA_CONV1_LOAD(nxt, 2 * tp + 2)
            }
            MRCA_PIN();
            c1a[cur] = zero16();
            c1b[cur] = zero16();
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                c1a[cur] = MRCA_MFMA(a1[s], ba[cur][s], c1a[cur]);
                c1b[cur] = MRCA_MFMA(a1[s], bb[cur][s], c1b[cur]);
                if (tp > 0) {
                    MRCA_CONV1_OUT(tp - 1, 2 * s)
                    MRCA_CONV1_OUT(tp - 1, 2 * s + 1)
                }
                // the previous robot's pair 1 (for the first robot: zeros through the same LDS columns, nothing stored)
                if (tp == 0) {
                    MRCA_CONV2_OUT(p1a, p1b, 1, 2 * s, 2)
                }
                if (tp == 1) row[s] = *reinterpret_cast<const f32x4*>(orow + 4 * s * kHPitch + 64);
                if (tp == 2 && have_prev) *reinterpret_cast<f32x4*>(out_prev + 4 * s * kL2 + 64) = row[s];
                if (tp == 3 && s == 5) {
                    MRCA_CONV2_LOAD(0, 0)      // conv2's first operands: H1 positions <= 127, written two pairs ago
                }
                MRCA_PIN();
            }
            MRCA_FSTAMP(tp)
        }

        // --- conv2: two pairs of position tiles (positions 64 P .. 64 P + 63), 12 chunks of four K steps each
        f32x16 acc0, acc1, done0, done1;
#pragma unroll
        for (int g = 0; g < 24; ++g) {
            const int P = g / 12, ch = g % 12;
            const int cur = g & 1, nxt = cur ^ 1;
            if (ch == 0) {
                acc0 = splat16(bias2);
                acc1 = splat16(bias2);
            }
            // four K steps; what rides along is dealt out over them so that nothing but an MFMA pair is ever longer than
            // the 128 cycles the pair keeps the matrix pipe busy
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int s = 4 * ch + k;
                acc0 = MRCA_MFMA(a2[s], b0[cur][k], acc0);
                acc1 = MRCA_MFMA(a2[s], b1v[cur][k], acc1);
                if (g < 23) {
                    MRCA_CONV2_LOAD_K(nxt, g + 1, k)
                }
                if (P == 0) {
                    if (ch < 8) {                    // conv1 pair 3 (conv2 pair 0 reads nothing of it)
                        if (k < 2) {
                            MRCA_CONV1_OUT(3, 2 * ch + k)
                        }
                    } else if (ch < 11) {            // the next robot's scan (its loads were requested a whole robot ago)
                        if (k < 2 && n + stride < n_robots) {
                            MRCA_STAGE_SCAN(2 * (ch - 8) + k)
                        }
                        if (ch == 8 && k == 2) {
                            MRCA_H1_PAD()
                        }
                    } else if (k == 0) {
                        MRCA_REQUEST_NEXT_AFTER()
                    }
                } else {
                    // pair 0's output: accumulators -> H1E[c][0..63] (pair 1 reads columns >= 64 only), rows back as
                    // float4, out
                    if (ch < 4) {
                        MRCA_CONV2_OUT(done0, done1, 0, 4 * ch + k, 1)
                    } else if (ch < 8) {
                        if (k < 2) row[2 * (ch - 4) + k] = *reinterpret_cast<const f32x4*>(orow + 4 * (2 * (ch - 4) + k) * kHPitch);
                    } else {
                        if (k < 2) *reinterpret_cast<f32x4*>(out + 4 * (2 * (ch - 8) + k) * kL2) = row[2 * (ch - 8) + k];
                        if (ch == 11 && k == 2) {
                            MRCA_CONV1_LOAD(0, 0)   // the next robot's first conv1 operands (staged during pair 0)
                        }
                    }
                }
                MRCA_PIN();
            }
            if (g == 11) {
                done0 = acc0;
                done1 = acc1;
                MRCA_FSTAMP(4)
            }
            if (g == 23) {
                MRCA_FSTAMP(5)
            }
        }
        p1a = acc0;
        p1b = acc1;
        out_prev = out;
        have_prev = true;
#if defined(MRCA_PROFILING)
        fst[6] += 1;
#endif
    }
#if defined(MRCA_PROFILING)
    fst[7] = __builtin_amdgcn_s_memrealtime() - freal0;
    if (lane == 0 && gwave < kFwdStampWaves)
        for (int k = 0; k < kFwdStamps; ++k) g_fwd_stamps[k][gwave] = fst[k];
#endif
    // --- the last robot's pair 1 leaves
    MRCA_CONV2_OUT(p1a, p1b, 1, 0, 16)
    MRCA_PIN();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(orow + 4 * q * kHPitch + 64);
        *reinterpret_cast<f32x4*>(out_prev + 4 * q * kL2 + 64) = v;
    }
#undef MRCA_STAGE_SCAN
#undef MRCA_REQUEST_NEXT
#undef MRCA_REQUEST_NEXT_AFTER
#undef MRCA_CONV1_LOAD
#undef MRCA_CONV1_OUT
#undef MRCA_CONV2_LOAD
#undef MRCA_CONV2_LOAD_K
#undef MRCA_H1_PAD
#undef MRCA_CONV2_OUT
}

}  // namespace mrca_policy

namespace mrca_policy {
struct DeviceInfo {
    int cus = 0;
    bool attr_set = false;
};
static DeviceInfo g_dev[64];     // per DEVICE: CU count and the dynamic-LDS attribute (a second GPU needs its own)
}  // namespace mrca_policy

#if defined(MRCA_PROFILING)
// Profiling build only: where the waves of the LAST mrca_lidar_features launch spent their time.  out[0..3] = conv1 tile
// pairs 0..3 (pair 0 carries the previous robot's output along), out[4] / out[5] = conv2 tile pairs 0 / 1: s_memtime ticks
// per robot, averaged over the waves that had work; out[6] = robots per wave; out[7] = the shader clock during the loop
// [GHz] (s_memtime against the constant 100 MHz s_memrealtime).  Synchronises the device.
extern "C" int mrca_debug_fwd_stamps(double* out /* [8] */) {
    using namespace mrca_policy;
    if (!out) return mrca::set_error(MRCA_ERR_INVALID, "mrca_debug_fwd_stamps: NULL");
    if (hipDeviceSynchronize() != hipSuccess) return mrca::set_error(MRCA_ERR_HIP, "mrca_debug_fwd_stamps: sync failed");
    static unsigned long long h[kFwdStamps][kFwdStampWaves];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fwd_stamps), sizeof(h), 0, hipMemcpyDeviceToHost) != hipSuccess)
        return mrca::set_error(MRCA_ERR_HIP, "mrca_debug_fwd_stamps: copy failed");
    double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int waves = 0;
    for (int w = 0; w < kFwdStampWaves; ++w) {
        if (h[6][w] == 0) continue;
        ++waves;
        double ticks = 0.0;
        for (int k = 0; k < 6; ++k) {
            sum[k] += (double)h[k][w] / (double)h[6][w];
            ticks += (double)h[k][w];
        }
        sum[6] += (double)h[6][w];
        sum[7] += ticks / ((double)h[7][w] * 10.0);      // s_memtime ticks per ns of the 100 MHz counter = GHz
    }
    for (int k = 0; k < 8; ++k) out[k] = waves ? sum[k] / waves : 0.0;
    return MRCA_OK;
}
#endif

static int lidar_features_impl(const float* obs_dev, const uint8_t* obs_head_dev, const int32_t* rows_dev, int32_t raw_scans,
                               int32_t n_robots, int32_t frames, int32_t beams, const float* w1_dev, const float* b1_dev,
                               const float* w2_dev, const float* b2_dev, float* feat_dev, void* stream) {
    using namespace mrca_policy;
    if (!obs_dev || !w1_dev || !b1_dev || !w2_dev || !b2_dev || !feat_dev)
        return mrca::set_error(MRCA_ERR_INVALID, "mrca_lidar_features: NULL pointer");
    if (frames != kFrames || beams != kBeams || n_robots < 1)
        return mrca::set_error(MRCA_ERR_UNSUPPORTED, "mrca_lidar_features: frames %d beams %d robots %d (needs 3 x 512, >= 1)",
                               frames, beams, n_robots);
    mrca::DeviceGuard guard(mrca::device_of(obs_dev));     // launch where the buffers live
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return mrca::set_error(MRCA_ERR_HIP, "mrca_lidar_features: hipGetDevice failed");
    DeviceInfo& d = g_dev[dev];
    const size_t lds = (size_t)kWavesPerBlock * kWaveFloats * sizeof(float);   // 160 128 B
    if (d.cus == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        d.cus = cus;
    }
    if (!d.attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lidar_features_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(lidar_features_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return mrca::set_error(MRCA_ERR_HIP, "mrca_lidar_features: %zu B of dynamic LDS refused: %s", lds,
                                   hipGetErrorString(e));
        d.attr_set = true;
    }
    // persistent waves: one workgroup of 4 waves per CU (156 kB of LDS), each wave pair walks every (#pairs)-th robot
    int blocks = d.cus;
    const int pairs_needed = (n_robots + 1) / 2;          // a block holds two (actor, critic) pairs
    if (blocks > pairs_needed) blocks = pairs_needed;
    if (raw_scans)
        hipLaunchKernelGGL(lidar_features_kernel<true>, dim3(blocks), dim3(64 * kWavesPerBlock), lds,
                           static_cast<hipStream_t>(stream), obs_dev, obs_head_dev, rows_dev, n_robots, w1_dev, b1_dev, w2_dev, b2_dev, feat_dev);
    else
        hipLaunchKernelGGL(lidar_features_kernel<false>, dim3(blocks), dim3(64 * kWavesPerBlock), lds,
                           static_cast<hipStream_t>(stream), obs_dev, obs_head_dev, rows_dev, n_robots, w1_dev, b1_dev, w2_dev, b2_dev, feat_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mrca::set_error(MRCA_ERR_HIP, "mrca_lidar_features launch: %s", hipGetErrorString(e));
    return MRCA_OK;
}

extern "C" int mrca_lidar_features(const float* obs_dev, const uint8_t* obs_head_dev, int32_t raw_scans, int32_t n_robots,
                                   int32_t frames, int32_t beams, const float* w1_dev, const float* b1_dev, const float* w2_dev, const float* b2_dev,
                                   float* feat_dev, void* stream) {
    return lidar_features_impl(obs_dev, obs_head_dev, nullptr, raw_scans, n_robots, frames, beams, w1_dev, b1_dev, w2_dev, b2_dev, feat_dev,
                               stream);
}

extern "C" int mrca_lidar_features_rows(const float* frames_dev, const int32_t* rows_dev, int32_t n_samples, int32_t frames, int32_t beams,
                                        const float* w1_dev, const float* b1_dev, const float* w2_dev, const float* b2_dev,
                                        float* feat_dev, void* stream) {
    if (!rows_dev) return mrca::set_error(MRCA_ERR_INVALID, "mrca_lidar_features_rows: rows_dev is NULL");
    return lidar_features_impl(frames_dev, nullptr, rows_dev, 0, n_samples, frames, beams, w1_dev, b1_dev, w2_dev, b2_dev, feat_dev, stream);
}
