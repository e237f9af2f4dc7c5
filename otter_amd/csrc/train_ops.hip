// fp32 kernels of the training step: everything downstream of the frozen CLIP towers is trainable
// (otter/policy/otter.py:86-128; 31 M parameters) and is differentiated by hand here.
//
// Sizes are small (tokens per GPU-step = batch x seq_length = a few hundred rows, widths <= 2304), so the kernels
// favour generality and determinism over peak rate: no atomics anywhere (every output element has exactly one
// owner, reductions are two-stage), arbitrary strides, bounds-checked edges, and "accumulate" variants (+=) so that
// gradients of a tensor consumed twice add up without a separate add kernel.
#include "common.hpp"
#include "otter_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------
// C[M,N] (+)= op(A)[M,K] . op(B)[K,N] (+ bias[n]),  A(m,k) = A[m*sam + k*sak],  B(k,n) = B[k*sbk + n*sbn].
// One workgroup (4 waves) = one 64x64 or 32x32 output tile on v_mfma_f32_16x16x4_f32, K-steps of 64 through LDS with the
// operands of the next two steps in flight in registers (see sgemm_strided_kernel).
// Used for the forward (x W^T), the input gradient (dy W) and the weight gradient (dy^T x) of every linear layer,
// straight from the parameter storage (no packed copies to refresh after an optimizer step).
struct SGemm {
    const float* A; long sam, sak;
    const float* B; long sbk, sbn;
    const float* bias;
    float* C; long ldc;
    int M, N, K, accumulate;
    float* a_rowsum;            // [M] or null: (+)= sum_k A(m, k) -- the bias gradient beside a weight gradient dy^T . x
};

constexpr int SG_K = 64;

// The problems of the trainable tail are small (192 tokens x 768 columns), and fp32 MFMA gives a CU only ~0.6 TFLOP/s, so a
// launch must spread over as many CUs as it can: T = 32 (one 16x16 MFMA tile per wave) when 64x64 tiles would number fewer
// than two per CU, T = 64 (2x2 per wave) otherwise.  A launch is then a short chain of K-steps:
//  * two K-steps of 64 are kept in flight in registers (the loads of K-step s+2 are requested as soon as K-step s has been
//    copied to LDS), so no K-step waits for memory;
//  * an operand tile lies in LDS with the index that is contiguous in MEMORY contiguous in LDS too -- [m][k] (pitch LDK) for
//    a k-contiguous operand (AK / !BN), [k][m] (pitch LD) for an m-contiguous one: the threads copy consecutive memory
//    elements to consecutive banks, and both pitches are 4 mod 32, which keeps the MFMA operand reads (16 lanes along one
//    index, 4 along the other) conflict-free in either layout.  The layouts are template parameters, so every LDS offset of
//    the product phase is an immediate;
//  * VEC: 16-byte loads and LDS writes along the contiguous index when strides, extents and pointers allow (4x fewer
//    instructions in the two copy phases, which were half of a K-step);
//  * the product phase first requests all fragments of the K-step, then issues its MFMAs back to back;
//  * a_rowsum: the workgroups of the first column tile also add up the rows of their A tiles from LDS (T/4 elements per
//    thread and K-step), which saves the weight-gradient call of a linear layer its separate column-sum launch for the bias.
template <int T, bool AK, bool BN, bool VEC>
__global__ __launch_bounds__(256) void sgemm_strided_kernel(SGemm p) {
    constexpr int LD = T + 4, LDK = SG_K + 4, R = T / 32;
    constexpr int G = VEC ? 4 : 1;                                  // elements per copy instruction
    constexpr int E = T * SG_K / 256 / G;                           // copy instructions per operand, thread and K-step
    constexpr int TILE_F = (SG_K * LD > T * LDK) ? SG_K * LD : T * LDK;
    typedef float cp_t __attribute__((ext_vector_type(G)));
    __shared__ __attribute__((aligned(16))) float As[TILE_F];
    __shared__ __attribute__((aligned(16))) float Bs[TILE_F];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int wm = wave & 1, wn = wave >> 1;
    const int m0 = blockIdx.y * T, n0 = blockIdx.x * T;

    f32x4 acc[R][R];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < R; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // copy instruction i of this thread: G consecutive elements along the operand's contiguous index.  Rows / columns past
    // the edge read a clamped address (they only feed outputs that are never stored).
    int ka[E], kb[E], lds_a[E], lds_b[E];
    long off_a[E], off_b[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int e = tid + 256 * i;
        int k, m, k2, n;
        if (AK) { k = (e % (SG_K / G)) * G; m = e / (SG_K / G); } else { m = (e % (T / G)) * G; k = e / (T / G); }
        if (BN) { n = (e % (T / G)) * G; k2 = e / (T / G); } else { k2 = (e % (SG_K / G)) * G; n = e / (SG_K / G); }
        ka[i] = k; kb[i] = k2;
        lds_a[i] = AK ? m * LDK + k : k * LD + m;
        lds_b[i] = BN ? k2 * LD + n : n * LDK + k2;
        off_a[i] = (long)min(m0 + m, p.M - (AK ? 1 : G)) * p.sam + (long)k * p.sak;
        off_b[i] = (long)k2 * p.sbk + (long)min(n0 + n, p.N - (BN ? G : 1)) * p.sbn;
    }
    // MFMA operand reads: A element (k = 4 kk + g, m = this wave's rows + 16 i + li), B likewise with n
    const int rd_a = AK ? (wm * (T / 2) + li) * LDK + g : g * LD + wm * (T / 2) + li;
    const int rd_b = BN ? g * LD + wn * (T / 2) + li : (wn * (T / 2) + li) * LDK + g;
    constexpr int A_KK = AK ? 4 : 4 * LD, A_I = AK ? 16 * LDK : 16;       // offsets per kk and per 16-row tile
    constexpr int B_KK = BN ? 4 * LD : 4, B_J = BN ? 16 : 16 * LDK;

    const bool do_rs = p.a_rowsum != nullptr && blockIdx.x == 0;       // (uniform in the workgroup)
    const int rs_m = tid % T, rs_kq = tid / T;                          // row and K slice (of 256 / T) this thread adds up
    float rs = 0.f;
    cp_t av[2][E], bv[2][E];              // two K-steps in flight
    // Every load is unconditional (a load inside a branch would make the compiler drain the whole queue at its first use, and
    // with it the K-step requested behind): in the ragged last K-step (and in the dummy requests past it, which keep the
    // request count the same on every path) k is clamped.
    auto fetch = [&](cp_t (&a)[E], cp_t (&b)[E], int k0) {
        const float* Ak = p.A + (long)k0 * p.sak;
        const float* Bk = p.B + (long)k0 * p.sbk;
        if (k0 + SG_K <= p.K) {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                a[i] = *(const cp_t*)(Ak + off_a[i]);
                b[i] = *(const cp_t*)(Bk + off_b[i]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                a[i] = *(const cp_t*)(Ak + off_a[i] + (long)(min(k0 + ka[i], p.K - (AK ? G : 1)) - (k0 + ka[i])) * p.sak);
                b[i] = *(const cp_t*)(Bk + off_b[i] + (long)(min(k0 + kb[i], p.K - (BN ? 1 : G)) - (k0 + kb[i])) * p.sbk);
            }
        }
    };
    auto k_step = [&](cp_t (&a)[E], cp_t (&b)[E], int k0, int k_next) {
        __syncthreads();                  // previous K-step's fragments have been read
        if (k0 + SG_K <= p.K) {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                *(cp_t*)(As + lds_a[i]) = a[i];
                *(cp_t*)(Bs + lds_b[i]) = b[i];
            }
        } else {                          // elements past K must be zero (a VEC group along k lies wholly inside or outside: K % 4 == 0)
#pragma unroll
            for (int i = 0; i < E; ++i) {
                *(cp_t*)(As + lds_a[i]) = k0 + ka[i] < p.K ? a[i] : (cp_t)(0.f);
                *(cp_t*)(Bs + lds_b[i]) = k0 + kb[i] < p.K ? b[i] : (cp_t)(0.f);
            }
        }
        __syncthreads();
        fetch(a, b, k_next);              // two K-steps ahead, into the registers just copied out
        if (do_rs) {
#pragma unroll
            for (int j = 0; j < T / 4; ++j) {
                const int k = rs_kq * (T / 4) + j;
                rs += AK ? As[rs_m * LDK + k] : As[k * LD + rs_m];
            }
        }
        float fa[SG_K / 4][R], fb[SG_K / 4][R];
#pragma unroll
        for (int kk = 0; kk < SG_K / 4; ++kk) {
#pragma unroll
            for (int i = 0; i < R; ++i) fa[kk][i] = As[rd_a + kk * A_KK + i * A_I];
#pragma unroll
            for (int j = 0; j < R; ++j) fb[kk][j] = Bs[rd_b + kk * B_KK + j * B_J];
        }
#pragma unroll
        for (int kk = 0; kk < SG_K / 4; ++kk)
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int j = 0; j < R; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
    };
    if (p.K > 0) {
        fetch(av[0], bv[0], 0);
        fetch(av[1], bv[1], SG_K);
    }
    for (int k0 = 0; k0 < p.K; k0 += 2 * SG_K) {
        k_step(av[0], bv[0], k0, k0 + 2 * SG_K);
        if (k0 + SG_K < p.K) k_step(av[1], bv[1], k0 + SG_K, k0 + 3 * SG_K);
    }
    if (do_rs) {
        __syncthreads();                  // the last K-step's fragments have been read: As is free
        As[rs_kq * T + rs_m] = rs;
        __syncthreads();
        if (tid < T && m0 + tid < p.M) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 256 / T; ++q) t += As[q * T + tid];
            float* dst = p.a_rowsum + m0 + tid;
            *dst = p.accumulate ? *dst + t : t;
        }
    }
    // D layout: column (n) = lane & 15, row (m) = 4 * (lane >> 4) + r
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int n = n0 + wn * (T / 2) + j * 16 + li;
            if (n >= p.N) continue;
            const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * (T / 2) + i * 16 + 4 * g + r;
                if (m >= p.M) continue;
                float* c = p.C + (long)m * p.ldc + n;
                const float v = acc[i][j][r] + bias;
                *c = p.accumulate ? *c + v : v;
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------
// activations: y = act(z);  dz += dy * act'(z)
__device__ __forceinline__ float act_eval(float z, int act) {
    if (act == OTTER_EPI_ACT_RELU) return fmaxf(z, 0.f);
    if (act == OTTER_EPI_ACT_GELU) return gelu_erf(z);
    if (act == OTTER_EPI_ACT_QUICKGELU) return quick_gelu(z);
    return z;
}
__device__ __forceinline__ float act_grad(float z, int act) {
    if (act == OTTER_EPI_ACT_RELU) return z > 0.f ? 1.f : 0.f;
    if (act == OTTER_EPI_ACT_GELU) {
        const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752f));
        return cdf + z * 0.39894228040143268f * __expf(-0.5f * z * z);
    }
    if (act == OTTER_EPI_ACT_QUICKGELU) {
        const float s = 1.0f / (1.0f + __expf(-1.702f * z));
        return s + 1.702f * z * s * (1.0f - s);
    }
    return 1.f;
}
__global__ void act_fwd_kernel(const float* z, float* y, long n, int act) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = act_eval(z[i], act);
}
__global__ void act_bwd_kernel(const float* z, const float* dy, float* dz, long n, int act) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dz[i] += dy[i] * act_grad(z[i], act);
}

// ------------------------------------------------------------------------------------------------------------------
// out[c] (+)= sum_r x[r*ld + c] * (y ? y[r*ldy + c] : 1):   bias gradients, broadcast-parameter gradients
__global__ __launch_bounds__(256) void col_sum_kernel(const float* x, long ldx, const float* y, long ldy, int rows, int cols,
                                                      float* out, int accumulate) {
    // a workgroup owns 16 columns: 16 row groups x 16 columns, combined through LDS in a fixed order
    __shared__ float part[16][17];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float s = 0.f;
    if (c < cols)
        for (int r = rg; r < rows; r += 16) s += x[(long)r * ldx + c] * (y ? y[(long)r * ldy + c] : 1.0f);
    part[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += part[i][cl];
        out[c] = accumulate ? out[c] + t : t;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// LayerNorm backward.  y = xh * gamma + beta, xh = (x - mean) * rstd.
//   dx += rstd * (g - mean(g) - xh * mean(g * xh)),  g = dy * gamma          one wave per row
//   stats[r] = (mean, rstd) for the parameter-gradient kernel
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* x, long ldx, const float* gamma, const float* dy, long ldy,
                                                     float* dx, long lddx, float* stats, int rows, int width, float eps) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* xr = x + (long)r * ldx;
    const float* dyr = dy + (long)r * ldy;
    float s = 0.f;
    for (int i = lane; i < width; i += 64) s += xr[i];
    const float mean = wave_sum(s) / (float)width;
    float q = 0.f;
    for (int i = lane; i < width; i += 64) { const float t = xr[i] - mean; q += t * t; }
    const float rstd = rsqrtf(wave_sum(q) / (float)width + eps);
    float sg = 0.f, sgx = 0.f;
    for (int i = lane; i < width; i += 64) {
        const float gv = dyr[i] * (gamma ? gamma[i] : 1.0f);
        sg += gv;
        sgx += gv * (xr[i] - mean) * rstd;
    }
    sg = wave_sum(sg) / (float)width;
    sgx = wave_sum(sgx) / (float)width;
    if (dx) {
        float* dxr = dx + (long)r * lddx;
        for (int i = lane; i < width; i += 64) {
            const float gv = dyr[i] * (gamma ? gamma[i] : 1.0f);
            dxr[i] += rstd * (gv - sg - (xr[i] - mean) * rstd * sgx);
        }
    }
    if (stats && lane == 0) { stats[2 * r] = mean; stats[2 * r + 1] = rstd; }
}
// dgamma[c] += sum_r dy * xh ; dbeta[c] += sum_r dy
__global__ __launch_bounds__(256) void ln_param_grad_kernel(const float* x, long ldx, const float* dy, long ldy, const float* stats,
                                                            int rows, int width, float* dgamma, float* dbeta) {
    __shared__ float pg[16][17], pb[16][17];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float sg = 0.f, sb = 0.f;
    if (c < width) {
        // four rows per pass, their loads requested together: one row per pass made the loop a chain of memory round trips
        int r = rg;
        for (; r + 48 < rows; r += 64) {
            float d[4], xv[4], mu[4], rs[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ru = r + 16 * u;
                d[u] = dy[(long)ru * ldy + c];
                xv[u] = x[(long)ru * ldx + c];
                mu[u] = stats[2 * ru];
                rs[u] = stats[2 * ru + 1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                sg += d[u] * (xv[u] - mu[u]) * rs[u];
                sb += d[u];
            }
        }
        for (; r < rows; r += 16) {
            const float d = dy[(long)r * ldy + c];
            sg += d * (x[(long)r * ldx + c] - stats[2 * r]) * stats[2 * r + 1];
            sb += d;
        }
    }
    pg[rg][cl] = sg;
    pb[rg][cl] = sb;
    __syncthreads();
    if (rg == 0 && c < width) {
        float tg = 0.f, tb = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { tg += pg[i][cl]; tb += pb[i][cl]; }
        dgamma[c] += tg;
        dbeta[c] += tb;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Small multi-head attention, forward and backward (policy: causal, 12-32 positions, head_dim 96; pooling readouts:
// 4 queries x 15 keys with a key-padding mask).  One workgroup per (batch, head); everything in LDS.
// Element (b, i, h, e) of q is q[b*sqb + i*sql + h*hd + e]; likewise k, v (skb, skl), o / do (sob, sol).
// q_bstride 0 broadcasts one query set over the batch (the learned readout queries).
struct SmallAttn {
    const float* q; long sqb, sql;
    const float* k; const float* v; long skb, skl;
    const unsigned char* kmask; long smb;       // [B][Lk], 1 = ignore key; may be null
    float* o; long sob, sol;                    // forward output / backward dO (const use)
    float* dq; float* dk; float* dv;            // backward outputs (+=); dq strides as q (sqb may not be 0), dk/dv as k
    long sdqb, sdql;
    int B, H, Lq, Lk, hd, causal;
    float scale;
};
constexpr int SA_MAXL = 32, SA_MAXD = 128;

__device__ __forceinline__ void sa_load(const SmallAttn& p, int b, int h, float* qs, float* ks, float* vs) {
    for (int idx = threadIdx.x; idx < p.Lq * p.hd; idx += 256) {
        const int i = idx / p.hd, e = idx % p.hd;
        qs[i * SA_MAXD + e] = p.q[(long)b * p.sqb + (long)i * p.sql + h * p.hd + e];
    }
    for (int idx = threadIdx.x; idx < p.Lk * p.hd; idx += 256) {
        const int j = idx / p.hd, e = idx % p.hd;
        const long off = (long)b * p.skb + (long)j * p.skl + h * p.hd + e;
        ks[j * SA_MAXD + e] = p.k[off];
        vs[j * SA_MAXD + e] = p.v[off];
    }
}
__device__ __forceinline__ void sa_probs(const SmallAttn& p, int b, const float* qs, const float* ks, float* ps) {
    for (int idx = threadIdx.x; idx < p.Lq * p.Lk; idx += 256) {
        const int i = idx / p.Lk, j = idx % p.Lk;
        float acc = 0.f;
        for (int e = 0; e < p.hd; ++e) acc += qs[i * SA_MAXD + e] * ks[j * SA_MAXD + e];
        acc *= p.scale;
        if ((p.causal && j > i) || (p.kmask && p.kmask[(long)b * p.smb + j])) acc = -INFINITY;
        ps[i * SA_MAXL + j] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < p.Lq; i += 256) {
        float mx = -INFINITY;
        for (int j = 0; j < p.Lk; ++j) mx = fmaxf(mx, ps[i * SA_MAXL + j]);
        float sum = 0.f;
        for (int j = 0; j < p.Lk; ++j) { const float e = expf(ps[i * SA_MAXL + j] - mx); ps[i * SA_MAXL + j] = e; sum += e; }
        const float inv = 1.0f / sum;
        for (int j = 0; j < p.Lk; ++j) ps[i * SA_MAXL + j] *= inv;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void small_attn_fwd_kernel(SmallAttn p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* qs = sm; float* ks = qs + SA_MAXL * SA_MAXD; float* vs = ks + SA_MAXL * SA_MAXD; float* ps = vs + SA_MAXL * SA_MAXD;
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    sa_load(p, b, h, qs, ks, vs);
    __syncthreads();
    sa_probs(p, b, qs, ks, ps);
    for (int idx = threadIdx.x; idx < p.Lq * p.hd; idx += 256) {
        const int i = idx / p.hd, e = idx % p.hd;
        float acc = 0.f;
        for (int j = 0; j < p.Lk; ++j) acc += ps[i * SA_MAXL + j] * vs[j * SA_MAXD + e];
        p.o[(long)b * p.sob + (long)i * p.sol + h * p.hd + e] = acc;
    }
}

__global__ __launch_bounds__(256) void small_attn_bwd_kernel(SmallAttn p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* qs = sm; float* ks = qs + SA_MAXL * SA_MAXD; float* vs = ks + SA_MAXL * SA_MAXD; float* ps = vs + SA_MAXL * SA_MAXD;
    float* dos = ps + SA_MAXL * SA_MAXL;           // [Lq][hd]
    float* dss = dos + SA_MAXL * SA_MAXD;          // [Lq][Lk]
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    sa_load(p, b, h, qs, ks, vs);
    for (int idx = threadIdx.x; idx < p.Lq * p.hd; idx += 256) {
        const int i = idx / p.hd, e = idx % p.hd;
        dos[i * SA_MAXD + e] = p.o[(long)b * p.sob + (long)i * p.sol + h * p.hd + e];
    }
    __syncthreads();
    sa_probs(p, b, qs, ks, ps);
    // dP = dO V^T
    for (int idx = threadIdx.x; idx < p.Lq * p.Lk; idx += 256) {
        const int i = idx / p.Lk, j = idx % p.Lk;
        float acc = 0.f;
        for (int e = 0; e < p.hd; ++e) acc += dos[i * SA_MAXD + e] * vs[j * SA_MAXD + e];
        dss[i * SA_MAXL + j] = acc;
    }
    __syncthreads();
    // dS = P * (dP - rowsum(P * dP)) * scale
    for (int i = threadIdx.x; i < p.Lq; i += 256) {
        float dsum = 0.f;
        for (int j = 0; j < p.Lk; ++j) dsum += ps[i * SA_MAXL + j] * dss[i * SA_MAXL + j];
        for (int j = 0; j < p.Lk; ++j) dss[i * SA_MAXL + j] = ps[i * SA_MAXL + j] * (dss[i * SA_MAXL + j] - dsum) * p.scale;
    }
    __syncthreads();
    if (p.dq)
        for (int idx = threadIdx.x; idx < p.Lq * p.hd; idx += 256) {
            const int i = idx / p.hd, e = idx % p.hd;
            float acc = 0.f;
            for (int j = 0; j < p.Lk; ++j) acc += dss[i * SA_MAXL + j] * ks[j * SA_MAXD + e];
            p.dq[(long)b * p.sdqb + (long)i * p.sdql + h * p.hd + e] += acc;
        }
    for (int idx = threadIdx.x; idx < p.Lk * p.hd; idx += 256) {
        const int j = idx / p.hd, e = idx % p.hd;
        float ak = 0.f, av = 0.f;
        for (int i = 0; i < p.Lq; ++i) {
            ak += dss[i * SA_MAXL + j] * qs[i * SA_MAXD + e];
            av += ps[i * SA_MAXL + j] * dos[i * SA_MAXD + e];
        }
        const long off = (long)b * p.skb + (long)j * p.skl + h * p.hd + e;
        if (p.dk) p.dk[off] += ak;
        if (p.dv) p.dv[off] += av;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// loss = mean |pred - gt| (L1) or mean (pred - gt)^2 (MSE);  dpred = grad_scale * d loss / d pred.  One workgroup:
// a fixed summation order, so the loss value is reproducible.
__global__ __launch_bounds__(1024) void loss_kernel(const float* pred, const float* gt, long n, int mse, float grad_scale,
                                                    float* loss, float* dpred) {
    __shared__ float part[16];
    float s = 0.f;
    const float inv_n = 1.0f / (float)n;
    for (long i = threadIdx.x; i < n; i += 1024) {
        const float d = pred[i] - gt[i];
        s += mse ? d * d : fabsf(d);
        if (dpred) dpred[i] = grad_scale * inv_n * (mse ? 2.0f * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)));
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += part[i];
        *loss = t * inv_n;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// gradient statistics: out[0] = sum x^2, out[1] = number of non-finite elements (two-stage, fixed order)
constexpr int STAT_BLOCKS = 512;
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* x, long n, float* ws) {
    __shared__ float ps[4], pf[4];
    float s = 0.f, bad = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)STAT_BLOCKS * 256) {
        const float v = x[i];
        if (isfinite(v)) s += v * v; else bad += 1.f;
    }
    s = wave_sum(s); bad = wave_sum(bad);
    if ((threadIdx.x & 63) == 0) { ps[threadIdx.x >> 6] = s; pf[threadIdx.x >> 6] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ws[2 * blockIdx.x] = (ps[0] + ps[1]) + (ps[2] + ps[3]);
        ws[2 * blockIdx.x + 1] = (pf[0] + pf[1]) + (pf[2] + pf[3]);
    }
}
__global__ __launch_bounds__(64) void sumsq_final_kernel(const float* ws, float* out) {
    float s = 0.f, bad = 0.f;
    for (int i = threadIdx.x; i < STAT_BLOCKS; i += 64) { s += ws[2 * i]; bad += ws[2 * i + 1]; }
    s = wave_sum(s); bad = wave_sum(bad);
    if (threadIdx.x == 0) { out[0] = s; out[1] = bad; }
}

// ------------------------------------------------------------------------------------------------------------------
// AdamW (torch.optim.AdamW semantics, train.py:170-171): decoupled weight decay, bias-corrected moments.
// `skip` points at the non-finite count of the gradient statistics: when it is non-zero the step is skipped, as
// GradScaler.step does (misc.py:334-348).
__global__ void adamw_kernel(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                             float wd, float bc1, float bc2_sqrt, float inv_scale, const float* skip) {
    if (skip && *skip != 0.f) return;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gr = g[i] * inv_scale;
    const float mi = beta1 * m[i] + (1.0f - beta1) * gr;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gr * gr;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] * (1.0f - lr * wd) - (lr / bc1) * (mi / denom);
}

__global__ void scale_kernel(float* x, long n, float alpha) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= alpha;
}

// ------------------------------------------------------------------------------------------------------------------
// d loss / d temperature of the text-aware visual extraction (models.py:22-36), one workgroup per (observation, camera):
//   z_kp = s_kp / T,  a = softmax_p(z),  out_k = sum_p a_kp v_p,  v_p = xh_p + pos_p,  s_kp = th_k . xh_p
//   dz_kp = a_kp (c_kp - sum_q a_kq c_kq),  c_kp = g_k . v_p,     dT = -(1/T^2) sum_kp dz_kp s_kp
// partial[unit] receives the unit's contribution (summed per camera by col_sum afterwards).
struct TaveGrad {
    const void* patches; long unit_stride; int is_f16;
    const float* text;           // [n_text][K][D] un-normalised
    const float* pos;            // [cams][P][D] fp32
    const float* temperature;    // [cams]
    const float* gout;           // [cams][M][K][D]
    float* partial;              // [M][cams]
    int cams, P, D, K, t_rep, M;
};
template <typename T>
__global__ __launch_bounds__(256) void tave_temp_grad_kernel(TaveGrad p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int K = p.K, D = p.D, P = p.P;
    float* th = sm;                 // [K][D] normalised text
    float* gs = th + K * D;         // [K][D] upstream gradient
    float* sv = gs + K * D;         // [K][P] cosine similarities
    float* cv = sv + K * P;         // [K][P] g_k . v_p
    float* red = cv + K * P;        // [K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int unit = blockIdx.x, m = unit / p.cams, cam = unit % p.cams;
    const int b = m / p.t_rep;
    for (int k = wave; k < K; k += 4) {
        const float* src = p.text + ((long)b * K + k) * D;
        float sq = 0.f;
        for (int i = lane; i < D; i += 64) sq += src[i] * src[i];
        const float rn = 1.0f / sqrtf(wave_sum(sq));
        const float* gsrc = p.gout + (((long)cam * p.M + m) * K + k) * D;
        for (int i = lane; i < D; i += 64) { th[k * D + i] = src[i] * rn; gs[k * D + i] = gsrc[i]; }
    }
    __syncthreads();
    const T* X = (const T*)p.patches + (long)unit * p.unit_stride;
    const float* pos = p.pos + (long)cam * P * D;
    // A wave takes NP patches at a time: every text / gradient row read from LDS (16 bytes per lane) then serves NP dot
    // products -- one patch at a time the kernel re-read all 2 K D floats per patch and the LDS set its pace.
    constexpr int NP = 4, NV = 4;                              // D <= 1024: at most 4 groups of 4 elements of a row per lane
    typedef typename Half16<T>::v4 h4;
    const int nq = D / 4;                                      // 4-element groups per row
    for (int p0 = wave * NP; p0 < P; p0 += 4 * NP) {
        f32x4 xv[NP][NV], pv[NP][NV];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int pp = min(p0 + j, P - 1);                 // (patches past the end: a repeat of the last one, not stored)
            float sq = 0.f;
#pragma unroll
            for (int e = 0; e < NV; ++e) {
                const int q = e * 64 + lane;
                xv[j][e] = pv[j][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (q < nq) {
                    const h4 xr = *(const h4*)(X + (long)pp * D + 4 * q);
                    xv[j][e] = (f32x4){(float)xr[0], (float)xr[1], (float)xr[2], (float)xr[3]};
                    pv[j][e] = *(const f32x4*)(pos + (long)pp * D + 4 * q);
                    sq += xv[j][e][0] * xv[j][e][0] + xv[j][e][1] * xv[j][e][1] + xv[j][e][2] * xv[j][e][2] + xv[j][e][3] * xv[j][e][3];
                }
            }
            const float rn = 1.0f / sqrtf(wave_sum(sq));
#pragma unroll
            for (int e = 0; e < NV; ++e) { xv[j][e] *= rn; pv[j][e] += xv[j][e]; }          // xh and v = xh + pos
        }
        for (int k = 0; k < K; ++k) {
            f32x4 t4[NV], g4[NV];
#pragma unroll
            for (int e = 0; e < NV; ++e) {
                const int q = e * 64 + lane;
                t4[e] = g4[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (q < nq) {
                    t4[e] = *(const f32x4*)(th + k * D + 4 * q);
                    g4[e] = *(const f32x4*)(gs + k * D + 4 * q);
                }
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                float sk = 0.f, ck = 0.f;
#pragma unroll
                for (int e = 0; e < NV; ++e) {
                    sk += xv[j][e][0] * t4[e][0] + xv[j][e][1] * t4[e][1] + xv[j][e][2] * t4[e][2] + xv[j][e][3] * t4[e][3];
                    ck += pv[j][e][0] * g4[e][0] + pv[j][e][1] * g4[e][1] + pv[j][e][2] * g4[e][2] + pv[j][e][3] * g4[e][3];
                }
                sk = wave_sum(sk);
                ck = wave_sum(ck);
                if (lane == 0 && p0 + j < P) { sv[k * P + p0 + j] = sk; cv[k * P + p0 + j] = ck; }
            }
        }
    }
    __syncthreads();
    const float traw = p.temperature[cam];
    const float T_ = fminf(fmaxf(traw, 0.f), 100.f);
    for (int k = wave; k < K; k += 4) {
        float mx = -INFINITY;
        for (int i = lane; i < P; i += 64) mx = fmaxf(mx, sv[k * P + i] / T_);
        mx = wave_max(mx);
        float z = 0.f, ac = 0.f;
        for (int i = lane; i < P; i += 64) { const float e = expf(sv[k * P + i] / T_ - mx); z += e; ac += e * cv[k * P + i]; }
        z = wave_sum(z); ac = wave_sum(ac) / z;
        float acc = 0.f;
        for (int i = lane; i < P; i += 64) {
            const float a = expf(sv[k * P + i] / T_ - mx) / z;
            acc += a * (cv[k * P + i] - ac) * sv[k * P + i];
        }
        acc = wave_sum(acc);
        if (lane == 0) red[k] = acc;
    }
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int k = 0; k < K; ++k) t += red[k];
        const bool pass = traw >= 0.f && traw <= 100.f;          // torch.clamp passes the gradient inside [min, max]
        p.partial[(long)m * p.cams + cam] = pass ? -t / (T_ * T_) : 0.f;
    }
}

}  // namespace

// ==================================================================================================================
extern "C" int otter_sgemm_strided(const float* A, long sa_m, long sa_k, const float* B, long sb_k, long sb_n, const float* bias,
                                   float* C, long ldc, int M, int N, int K, int accumulate, float* a_rowsum, void* stream) {
    OTTER_REQUIRE(A && B && C, OTTER_ERR_INVALID_ARGUMENT, "otter_sgemm_strided: null operand");
    if (M == 0 || N == 0) return OTTER_OK;
    OTTER_REQUIRE(M > 0 && N > 0 && K >= 0 && ldc >= N, OTTER_ERR_INVALID_ARGUMENT, "otter_sgemm_strided: bad sizes M=%d N=%d K=%d ldc=%ld", M, N, K, ldc);
    SGemm a{A, sa_m, sa_k, B, sb_k, sb_n, bias, C, ldc, M, N, K, accumulate ? 1 : 0, a_rowsum};
    const bool small = (long)((N + 63) / 64) * ((M + 63) / 64) < 512;      // fewer than two 64x64 tiles per CU: 32x32 tiles (the fp32 MFMA rate of the busy CUs is the bound)
    const int t = small ? 32 : 64;
    dim3 grid((N + t - 1) / t, (M + t - 1) / t);
    OTTER_REQUIRE(grid.y < 65536, OTTER_ERR_UNSUPPORTED_SHAPE, "otter_sgemm_strided: M=%d too large", M);
    // layouts: which index of each operand is contiguous in memory (an operand with neither stride 1 takes the scalar
    // "m / n contiguous" form, which is correct for any strides)
    const bool ak = sa_k == 1, bn = !(sb_k == 1);
    auto mult4 = [](long v) { return (v & 3) == 0; };
    const bool vec = aligned16(A) && aligned16(B) && (ak ? mult4(sa_m) && mult4(K) : sa_m == 1 && mult4(sa_k) && mult4(M)) &&
                     (bn ? sb_n == 1 && mult4(sb_k) && mult4(N) : mult4(sb_n) && mult4(K));
    hipStream_t st = (hipStream_t)stream;
#define OTTER_SG_LAUNCH(T_, AK_, BN_, V_) hipLaunchKernelGGL((sgemm_strided_kernel<T_, AK_, BN_, V_>), grid, dim3(256), 0, st, a)
#define OTTER_SG_PICK_V(T_, AK_, BN_) do { if (vec) OTTER_SG_LAUNCH(T_, AK_, BN_, true); else OTTER_SG_LAUNCH(T_, AK_, BN_, false); } while (0)
#define OTTER_SG_PICK_L(T_) do { if (ak && bn) OTTER_SG_PICK_V(T_, true, true); else if (ak) OTTER_SG_PICK_V(T_, true, false); \
                                 else if (bn) OTTER_SG_PICK_V(T_, false, true); else OTTER_SG_PICK_V(T_, false, false); } while (0)
    if (small) OTTER_SG_PICK_L(32); else OTTER_SG_PICK_L(64);
#undef OTTER_SG_PICK_L
#undef OTTER_SG_PICK_V
#undef OTTER_SG_LAUNCH
    OTTER_CHECK_LAUNCH("otter_sgemm_strided");
    return OTTER_OK;
}

extern "C" int otter_act_fwd(const float* z, float* y, long n, int act, void* stream) {
    OTTER_REQUIRE(z && y && n >= 0, OTTER_ERR_INVALID_ARGUMENT, "otter_act_fwd: bad arguments");
    if (n == 0) return OTTER_OK;
    hipLaunchKernelGGL(act_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, y, n, act);
    OTTER_CHECK_LAUNCH("otter_act_fwd");
    return OTTER_OK;
}

extern "C" int otter_act_bwd(const float* z, const float* dy, float* dz, long n, int act, void* stream) {
    OTTER_REQUIRE(z && dy && dz && n >= 0, OTTER_ERR_INVALID_ARGUMENT, "otter_act_bwd: bad arguments");
    if (n == 0) return OTTER_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, dy, dz, n, act);
    OTTER_CHECK_LAUNCH("otter_act_bwd");
    return OTTER_OK;
}

extern "C" int otter_col_sum(const float* x, long ldx, const float* y, long ldy, int rows, int cols, float* out, int accumulate, void* stream) {
    OTTER_REQUIRE(x && out && rows >= 0 && cols > 0, OTTER_ERR_INVALID_ARGUMENT, "otter_col_sum: bad arguments");
    hipLaunchKernelGGL(col_sum_kernel, dim3((cols + 15) / 16), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, rows, cols, out, accumulate ? 1 : 0);
    OTTER_CHECK_LAUNCH("otter_col_sum");
    return OTTER_OK;
}

extern "C" int otter_layernorm_bwd(const float* x, long ldx, const float* gamma, const float* dy, long ldy, float* dx, long lddx,
                                   float* stats, float* dgamma, float* dbeta, int rows, int width, float eps, void* stream) {
    OTTER_REQUIRE(x && dy && rows >= 0 && width > 0, OTTER_ERR_INVALID_ARGUMENT, "otter_layernorm_bwd: bad arguments");
    OTTER_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), OTTER_ERR_INVALID_ARGUMENT, "otter_layernorm_bwd: dgamma and dbeta come together");
    OTTER_REQUIRE(!dgamma || stats, OTTER_ERR_INVALID_ARGUMENT, "otter_layernorm_bwd: parameter gradients need the stats workspace (2*rows floats)");
    if (rows == 0) return OTTER_OK;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ln_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, gamma, dy, ldy, dx, lddx, stats, rows, width, eps);
    OTTER_CHECK_LAUNCH("otter_layernorm_bwd");
    if (dgamma) {
        hipLaunchKernelGGL(ln_param_grad_kernel, dim3((width + 15) / 16), dim3(256), 0, s, x, ldx, dy, ldy, stats, rows, width, dgamma, dbeta);
        OTTER_CHECK_LAUNCH("otter_layernorm_bwd (parameters)");
    }
    return OTTER_OK;
}

static int small_attn_args_ok(int B, int H, int Lq, int Lk, int hd) {
    return B > 0 && H > 0 && Lq > 0 && Lk > 0 && hd > 0 && Lq <= SA_MAXL && Lk <= SA_MAXL && hd <= SA_MAXD && (long)B * H < (1L << 31);
}

extern "C" int otter_small_attn_fwd(const float* q, long sq_b, long sq_l, const float* k, const float* v, long sk_b, long sk_l,
                                    const unsigned char* key_mask, long sm_b, float* out, long so_b, long so_l, int B, int H, int Lq,
                                    int Lk, int head_dim, int causal, void* stream) {
    OTTER_REQUIRE(q && k && v && out, OTTER_ERR_INVALID_ARGUMENT, "otter_small_attn_fwd: null operand");
    if (B == 0) return OTTER_OK;
    OTTER_REQUIRE(small_attn_args_ok(B, H, Lq, Lk, head_dim), OTTER_ERR_UNSUPPORTED_SHAPE,
                  "otter_small_attn_fwd: B=%d H=%d Lq=%d Lk=%d hd=%d (lengths <= %d, head_dim <= %d)", B, H, Lq, Lk, head_dim, SA_MAXL, SA_MAXD);
    SmallAttn a{q, sq_b, sq_l, k, v, sk_b, sk_l, key_mask, sm_b, out, so_b, so_l, nullptr, nullptr, nullptr, 0, 0,
                B, H, Lq, Lk, head_dim, causal ? 1 : 0, 1.0f / sqrtf((float)head_dim)};
    const size_t lds = sizeof(float) * (3 * SA_MAXL * SA_MAXD + SA_MAXL * SA_MAXL);
    hipLaunchKernelGGL(small_attn_fwd_kernel, dim3(B * H), dim3(256), lds, (hipStream_t)stream, a);
    OTTER_CHECK_LAUNCH("otter_small_attn_fwd");
    return OTTER_OK;
}

extern "C" int otter_small_attn_bwd(const float* q, long sq_b, long sq_l, const float* k, const float* v, long sk_b, long sk_l,
                                    const unsigned char* key_mask, long sm_b, const float* dout, long so_b, long so_l,
                                    float* dq, long sdq_b, long sdq_l, float* dk, float* dv, int B, int H, int Lq, int Lk,
                                    int head_dim, int causal, void* stream) {
    OTTER_REQUIRE(q && k && v && dout, OTTER_ERR_INVALID_ARGUMENT, "otter_small_attn_bwd: null operand");
    if (B == 0) return OTTER_OK;
    OTTER_REQUIRE(small_attn_args_ok(B, H, Lq, Lk, head_dim), OTTER_ERR_UNSUPPORTED_SHAPE,
                  "otter_small_attn_bwd: B=%d H=%d Lq=%d Lk=%d hd=%d (lengths <= %d, head_dim <= %d)", B, H, Lq, Lk, head_dim, SA_MAXL, SA_MAXD);
    SmallAttn a{q, sq_b, sq_l, k, v, sk_b, sk_l, key_mask, sm_b, const_cast<float*>(dout), so_b, so_l, dq, dk, dv, sdq_b, sdq_l,
                B, H, Lq, Lk, head_dim, causal ? 1 : 0, 1.0f / sqrtf((float)head_dim)};
    const size_t lds = sizeof(float) * (4 * SA_MAXL * SA_MAXD + 2 * SA_MAXL * SA_MAXL);
    if (int rc = otter_reserve_lds((const void*)small_attn_bwd_kernel, lds, "otter_small_attn_bwd")) return rc;
    hipLaunchKernelGGL(small_attn_bwd_kernel, dim3(B * H), dim3(256), lds, (hipStream_t)stream, a);
    OTTER_CHECK_LAUNCH("otter_small_attn_bwd");
    return OTTER_OK;
}

extern "C" int otter_loss_fwd_bwd(const float* pred, const float* gt, long n, int mse, float grad_scale, float* loss, float* dpred, void* stream) {
    OTTER_REQUIRE(pred && gt && loss && n > 0, OTTER_ERR_INVALID_ARGUMENT, "otter_loss_fwd_bwd: bad arguments");
    hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, gt, n, mse ? 1 : 0, grad_scale, loss, dpred);
    OTTER_CHECK_LAUNCH("otter_loss_fwd_bwd");
    return OTTER_OK;
}

extern "C" long otter_grad_stats_workspace_floats(void) { return 2L * STAT_BLOCKS; }

extern "C" int otter_grad_stats(const float* x, long n, float* workspace, float* out2, void* stream) {
    OTTER_REQUIRE(x && workspace && out2 && n >= 0, OTTER_ERR_INVALID_ARGUMENT, "otter_grad_stats: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(STAT_BLOCKS), dim3(256), 0, s, x, n, workspace);
    OTTER_CHECK_LAUNCH("otter_grad_stats");
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, s, workspace, out2);
    OTTER_CHECK_LAUNCH("otter_grad_stats (final)");
    return OTTER_OK;
}

extern "C" int otter_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                                float beta2, float eps, float weight_decay, int step, float inv_scale, const float* skip_flag, void* stream) {
    OTTER_REQUIRE(param && grad && exp_avg && exp_avg_sq && n >= 0 && step >= 1, OTTER_ERR_INVALID_ARGUMENT, "otter_adamw_step: bad arguments");
    if (n == 0) return OTTER_OK;
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n,
                       lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, inv_scale, skip_flag);
    OTTER_CHECK_LAUNCH("otter_adamw_step");
    return OTTER_OK;
}

extern "C" int otter_scale_inplace(float* x, long n, float alpha, void* stream) {
    OTTER_REQUIRE(x && n >= 0, OTTER_ERR_INVALID_ARGUMENT, "otter_scale_inplace: bad arguments");
    if (n == 0) return OTTER_OK;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, alpha);
    OTTER_CHECK_LAUNCH("otter_scale_inplace");
    return OTTER_OK;
}

extern "C" int otter_text_aware_pool_temperature_grad(int dtype, const void* patches, long unit_stride, const float* text, const float* pos,
                                                      const float* temperature, const float* grad_out, float* partial, int n_obs, int cams,
                                                      int P, int D, int K, int t_rep, void* stream) {
    OTTER_REQUIRE(dtype == DT_BF16 || dtype == DT_F16, OTTER_ERR_INVALID_ARGUMENT, "otter_text_aware_pool_temperature_grad: dtype %d (bf16 or fp16)", dtype);
    OTTER_REQUIRE(patches && text && pos && temperature && grad_out && partial, OTTER_ERR_INVALID_ARGUMENT, "otter_text_aware_pool_temperature_grad: null operand");
    if (n_obs == 0) return OTTER_OK;
    OTTER_REQUIRE(n_obs > 0 && cams > 0 && t_rep > 0 && n_obs % t_rep == 0 && P > 0 && D > 0 && K > 0, OTTER_ERR_INVALID_ARGUMENT,
                  "otter_text_aware_pool_temperature_grad: bad sizes");
    OTTER_REQUIRE(D % 64 == 0 && D <= 1024, OTTER_ERR_UNSUPPORTED_SHAPE, "otter_text_aware_pool_temperature_grad: D=%d must be a multiple of 64, at most 1024", D);
    OTTER_REQUIRE(unit_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(patches) & 7u) == 0 && aligned16(pos), OTTER_ERR_ALIGNMENT,
                  "otter_text_aware_pool_temperature_grad: patches must be 8-byte aligned with unit_stride a multiple of 4, pos 16-byte aligned");
    const size_t lds = sizeof(float) * ((size_t)2 * K * D + (size_t)2 * K * P + K);
    OTTER_REQUIRE(lds <= 160 * 1024, OTTER_ERR_UNSUPPORTED_SHAPE, "otter_text_aware_pool_temperature_grad: needs %zu bytes of LDS", lds);
    TaveGrad a{patches, unit_stride, dtype == DT_F16, text, pos, temperature, grad_out, partial, cams, P, D, K, t_rep, n_obs};
    hipStream_t s = (hipStream_t)stream;
    const int which = dtype == DT_F16 ? 1 : 0;
    const void* fn = which ? (const void*)tave_temp_grad_kernel<_Float16> : (const void*)tave_temp_grad_kernel<__bf16>;
    if (int rc = otter_reserve_lds(fn, lds, "otter_text_aware_pool_temperature_grad")) return rc;
    if (which) hipLaunchKernelGGL(tave_temp_grad_kernel<_Float16>, dim3(n_obs * cams), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(tave_temp_grad_kernel<__bf16>, dim3(n_obs * cams), dim3(256), lds, s, a);
    OTTER_CHECK_LAUNCH("otter_text_aware_pool_temperature_grad");
    return OTTER_OK;
}
