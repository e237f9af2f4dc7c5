// Attention-pooling readout: the query side of the cross-attention blocks that compress (nk x D_in) tokens
// into R readout vectors of width d.  The key/value projections (the only part with real FLOPs) are done by
// the GEMM beforehand; this kernel does everything else of the pooling module in fp32, one workgroup
// per observation, with all intermediates and the block's weights in LDS:
//   per block:  qn = LN_q(q);  Q = qn Wq^T + bq;  scores = Q K^T / sqrt(hd) (+ key padding mask);
//               softmax over keys;  ctx = P V;  q1 = qn + ctx Wo^T + bo;      (residual on the NORMALISED query)
//               out = q1 + fc2(gelu(fc1(LN(q1))))
//   end:        LN(out) flattened to (R*d).
// Weights are passed transposed ([in][out]) so that consecutive threads read consecutive addresses.
// A launch is a dependent chain of ~25 barrier-separated phases per workgroup (~40 us whatever the batch), so the
// poolings of one encoder call -- one per camera and the text one, each with its own weights and sizes -- go into ONE
// launch: the grid is the concatenation of the problems' observations.
#include <algorithm>

#include "common.hpp"
#include "otter_hip.h"

namespace {

constexpr int MAX_BLOCKS = 4;

struct ReadoutArgs {
    otter_readout_block blk[MAX_BLOCKS];
    int n_blocks;
    const float* query;          // [R][d]
    const float* final_ln_w; const float* final_ln_b;
    const float* kv; long ldkv;  // rows (m*nk + k); block b: K at col b*2d, V at col b*2d + d
    const unsigned char* mask;   // [M][nk] (1 = ignore key) or null
    float* out; long ldo;        // [M][R*d]
    int M, nk, R, d, heads, hidden;
};

__device__ __forceinline__ void ln_row(const float* in, float* out, const float* w, const float* b, int d, int lane) {
    float s = 0.f;
    for (int i = lane; i < d; i += 64) s += in[i];
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
    for (int i = lane; i < d; i += 64) { const float t = in[i] - mean; q += t * t; }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + 1e-5f);
    for (int i = lane; i < d; i += 64) out[i] = (in[i] - mean) * rstd * w[i] + b[i];
}

// Bulk copy of n floats (global -> LDS) by the whole workgroup: whole KiB pieces by LDS-DMA (one wave-instruction
// each, nothing held in registers, all pieces in flight at once), the remainder by plain loads.  The caller
// synchronises (s_waitcnt vmcnt(0) + barrier, both in __syncthreads()) before reading dst.
__device__ __forceinline__ void stage_floats(float* dst, const float* src, int n, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int pieces = (((uintptr_t)src & 15) == 0) ? n / 256 : 0;          // 256 floats = 1 KiB
    for (int c = wave; c < pieces; c += 4)
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + c * 256 + lane * 4), LDS_PTR(dst + c * 256), 16, 0, 0);
    for (int i = pieces * 256 + tid; i < n; i += 256) dst[i] = src[i];
}

// Weights of a cross-attention block are shared by every observation and small (d*d and d*hidden floats), so each
// workgroup copies them into LDS in bulk -- one memory latency per matrix instead of one per inner-loop step --
// and the matrix-vector products then run at LDS speed.
constexpr int MAX_PROBLEMS = 4;
struct ReadoutMulti {
    ReadoutArgs a[MAX_PROBLEMS];
    int first[MAX_PROBLEMS + 1];      // problem i owns workgroups [first[i], first[i+1])
    int n;
};

__global__ __launch_bounds__(256) void readout_kernel(ReadoutMulti mp) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int pi = 0;
    while (pi + 1 < mp.n && (int)blockIdx.x >= mp.first[pi + 1]) ++pi;
    const ReadoutArgs& p = mp.a[pi];
    const int m = blockIdx.x - mp.first[pi];
    const int R = p.R, d = p.d, nk = p.nk, heads = p.heads, hd = d / heads, hidden = p.hidden;
    const int Rd = R * d;
    float* wS = sm;                   // [2][d][d]      Wq^T | Wo^T of the current block      (16-byte aligned: first)
    float* wF = wS + 2 * d * d;       // [d][hidden]    fc1^T, then fc2^T
    float* KV = wF + d * hidden;      // [nk][2d]
    float* q = KV + nk * 2 * d;       // [R][d] running query
    float* qn = q + Rd;               // [R][d]
    float* Q = qn + Rd;               // [R][d]
    float* q1 = Q + Rd;               // [R][d]
    float* hn = q1 + Rd;              // [R][d]
    float* sc = hn + Rd;              // [R][heads][nk]
    float* h1 = sc + R * heads * nk;  // [R][hidden]

    for (int i = tid; i < Rd; i += 256) q[i] = p.query[i];

    for (int b = 0; b < p.n_blocks; ++b) {
        const otter_readout_block& w = p.blk[b];
        // every weight of the block's attention half, fc1 and this observation's K | V rows: all loads in flight together
        stage_floats(wS, w.wq_t, d * d, tid);
        stage_floats(wS + d * d, w.wo_t, d * d, tid);
        stage_floats(wF, w.fc1_t, d * hidden, tid);
        for (int i = tid; i < nk * 2 * d; i += 256) {
            const int k = i / (2 * d), c = i % (2 * d);
            KV[i] = p.kv[((long)m * nk + k) * p.ldkv + b * 2 * d + c];
        }
        __syncthreads();                                    // q complete (previous iteration / initial copy)
        for (int r = wave; r < R; r += 4) ln_row(q + r * d, qn + r * d, w.q_ln_w, w.q_ln_b, d, lane);
        __syncthreads();
        for (int o = tid; o < Rd; o += 256) {
            const int r = o / d, j = o % d;
            float acc = w.bq[j];
#pragma unroll 8
            for (int i = 0; i < d; ++i) acc += qn[r * d + i] * wS[i * d + j];
            Q[o] = acc;
        }
        __syncthreads();
        const float scale = 1.0f / sqrtf((float)hd);
        for (int o = tid; o < R * heads * nk; o += 256) {
            const int k = o % nk, h = (o / nk) % heads, r = o / (nk * heads);
            float acc = 0.f;
            for (int e = 0; e < hd; ++e) acc += Q[r * d + h * hd + e] * KV[k * 2 * d + h * hd + e];
            acc *= scale;
            if (p.mask && p.mask[(long)m * nk + k]) acc = -INFINITY;
            sc[o] = acc;
        }
        __syncthreads();
        for (int o = tid; o < R * heads; o += 256) {
            float* row = sc + o * nk;
            float mx = -INFINITY;
            for (int k = 0; k < nk; ++k) mx = fmaxf(mx, row[k]);
            float sum = 0.f;
            for (int k = 0; k < nk; ++k) { const float e = expf(row[k] - mx); row[k] = e; sum += e; }
            const float inv = 1.0f / sum;
            for (int k = 0; k < nk; ++k) row[k] *= inv;
        }
        __syncthreads();
        for (int o = tid; o < Rd; o += 256) {
            const int r = o / d, j = o % d, h = j / hd;
            float acc = 0.f;
            for (int k = 0; k < nk; ++k) acc += sc[(r * heads + h) * nk + k] * KV[k * 2 * d + d + j];
            hn[o] = acc;                                  // hn used as ctx scratch here
        }
        __syncthreads();
        for (int o = tid; o < Rd; o += 256) {
            const int r = o / d, j = o % d;
            float acc = w.bo[j];
#pragma unroll 8
            for (int i = 0; i < d; ++i) acc += hn[r * d + i] * wS[d * d + i * d + j];
            q1[o] = qn[o] + acc;
        }
        __syncthreads();
        for (int r = wave; r < R; r += 4) ln_row(q1 + r * d, hn + r * d, w.ln_w, w.ln_b, d, lane);
        __syncthreads();
        for (int o = tid; o < R * hidden; o += 256) {
            const int r = o / hidden, u = o % hidden;
            float acc = w.fc1_b[u];
#pragma unroll 8
            for (int i = 0; i < d; ++i) acc += hn[r * d + i] * wF[i * hidden + u];
            h1[o] = gelu_erf(acc);
        }
        __syncthreads();                                    // fc1^T is dead: fc2^T takes its place
        stage_floats(wF, w.fc2_t, hidden * d, tid);
        __syncthreads();
        for (int o = tid; o < Rd; o += 256) {
            const int r = o / d, j = o % d;
            float acc = w.fc2_b[j];
#pragma unroll 8
            for (int u = 0; u < hidden; ++u) acc += h1[r * hidden + u] * wF[u * d + j];
            q[o] = q1[o] + acc;
        }
        __syncthreads();
    }
    for (int r = wave; r < R; r += 4) ln_row(q + r * d, p.out + (long)m * p.ldo + r * d, p.final_ln_w, p.final_ln_b, d, lane);
}

}  // namespace

static size_t readout_lds_bytes(int nk, int readouts, int d, int heads, int hidden) {
    return sizeof(float) * ((size_t)2 * d * d + (size_t)d * hidden + (size_t)5 * readouts * d + (size_t)nk * 2 * d +
                            (size_t)readouts * heads * nk + (size_t)readouts * hidden);
}

static int readout_check(const otter_readout_problem& q, int idx, ReadoutArgs& a) {
    OTTER_REQUIRE(q.blocks && q.query && q.final_ln_w && q.final_ln_b && q.kv && q.out, OTTER_ERR_INVALID_ARGUMENT,
                  "otter_readout_pool_fwd: problem %d has a null operand", idx);
    OTTER_REQUIRE(q.n_blocks >= 1 && q.n_blocks <= MAX_BLOCKS, OTTER_ERR_UNSUPPORTED_SHAPE, "otter_readout_pool_fwd: n_blocks=%d (1..%d)", q.n_blocks, MAX_BLOCKS);
    OTTER_REQUIRE(q.n_obs > 0 && q.nk > 0 && q.nk <= 64 && q.readouts > 0 && q.readouts <= 16 && q.d > 0 && q.d <= 256 && q.heads > 0 &&
                  q.d % q.heads == 0 && q.hidden > 0 && q.hidden <= 2048, OTTER_ERR_UNSUPPORTED_SHAPE,
                  "otter_readout_pool_fwd: unsupported sizes n_obs=%d nk=%d R=%d d=%d heads=%d hidden=%d", q.n_obs, q.nk, q.readouts, q.d, q.heads, q.hidden);
    OTTER_REQUIRE(q.ldkv >= (long)q.n_blocks * 2 * q.d && q.ldo >= (long)q.readouts * q.d, OTTER_ERR_INVALID_ARGUMENT,
                  "otter_readout_pool_fwd: ldkv=%ld / ldo=%ld too small", q.ldkv, q.ldo);
    a = ReadoutArgs{};
    for (int b = 0; b < q.n_blocks; ++b) {
        a.blk[b] = q.blocks[b];
        const otter_readout_block& w = q.blocks[b];
        OTTER_REQUIRE(w.q_ln_w && w.q_ln_b && w.wq_t && w.bq && w.wo_t && w.bo && w.ln_w && w.ln_b && w.fc1_t && w.fc1_b && w.fc2_t && w.fc2_b,
                      OTTER_ERR_INVALID_ARGUMENT, "otter_readout_pool_fwd: block %d has a null parameter", b);
    }
    a.n_blocks = q.n_blocks; a.query = q.query; a.final_ln_w = q.final_ln_w; a.final_ln_b = q.final_ln_b;
    a.kv = q.kv; a.ldkv = q.ldkv; a.mask = q.mask; a.out = q.out; a.ldo = q.ldo;
    a.M = q.n_obs; a.nk = q.nk; a.R = q.readouts; a.d = q.d; a.heads = q.heads; a.hidden = q.hidden;
    OTTER_REQUIRE(readout_lds_bytes(q.nk, q.readouts, q.d, q.heads, q.hidden) <= 160 * 1024, OTTER_ERR_UNSUPPORTED_SHAPE,
                  "otter_readout_pool_fwd: needs %zu bytes of LDS (> 160 KiB)", readout_lds_bytes(q.nk, q.readouts, q.d, q.heads, q.hidden));
    OTTER_REQUIRE((q.d * q.d) % 4 == 0 && (q.d * q.hidden) % 4 == 0, OTTER_ERR_UNSUPPORTED_SHAPE, "otter_readout_pool_fwd: d*d and d*hidden must be multiples of 4");
    return OTTER_OK;
}

extern "C" int otter_readout_pool_fwd_multi(const otter_readout_problem* problems, int n_problems, void* stream) {
    OTTER_REQUIRE(problems && n_problems >= 1 && n_problems <= MAX_PROBLEMS, OTTER_ERR_INVALID_ARGUMENT,
                  "otter_readout_pool_fwd_multi: %d problems (1..%d)", n_problems, MAX_PROBLEMS);
    ReadoutMulti mp{};
    size_t lds = 0;
    long total = 0;
    for (int i = 0; i < n_problems; ++i) {
        if (problems[i].n_obs == 0) continue;                // an empty problem takes no workgroups
        if (int rc = readout_check(problems[i], i, mp.a[mp.n])) return rc;
        lds = std::max(lds, readout_lds_bytes(problems[i].nk, problems[i].readouts, problems[i].d, problems[i].heads, problems[i].hidden));
        mp.first[mp.n] = (int)total;
        total += problems[i].n_obs;
        ++mp.n;
    }
    if (mp.n == 0) return OTTER_OK;
    OTTER_REQUIRE(total < (1L << 31), OTTER_ERR_UNSUPPORTED_SHAPE, "otter_readout_pool_fwd_multi: grid too large");
    for (int i = mp.n; i <= MAX_PROBLEMS; ++i) mp.first[i] = (int)total;
    if (int rc = otter_reserve_lds((const void*)readout_kernel, lds, "otter_readout_pool_fwd")) return rc;
    hipLaunchKernelGGL(readout_kernel, dim3((unsigned)total), dim3(256), lds, (hipStream_t)stream, mp);
    OTTER_CHECK_LAUNCH("otter_readout_pool_fwd");
    return OTTER_OK;
}

extern "C" int otter_readout_pool_fwd(const otter_readout_block* blocks, int n_blocks, const float* query,
                                      const float* final_ln_w, const float* final_ln_b, const float* kv, long ldkv,
                                      const unsigned char* mask, float* out, long ldo, int n_obs, int nk, int readouts,
                                      int d, int heads, int hidden, void* stream) {
    OTTER_REQUIRE(blocks && query && final_ln_w && final_ln_b && kv && out, OTTER_ERR_INVALID_ARGUMENT, "otter_readout_pool_fwd: null operand");
    if (n_obs == 0) return OTTER_OK;
    const otter_readout_problem q{blocks, n_blocks, query, final_ln_w, final_ln_b, kv, ldkv, mask, out, ldo, n_obs, nk, readouts, d, heads, hidden};
    return otter_readout_pool_fwd_multi(&q, 1, stream);
}
