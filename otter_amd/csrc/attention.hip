// Fused multi-head attention for the CLIP towers (head_dim 64) and the policy transformer.
//
// CLIP kernel (attn_d64): one workgroup (4 waves) per (sequence, head).  The sequence is short (257 vision
// tokens, <= 77 text tokens), so K and V of the head are staged once into LDS and each wave owns 16-query
// tiles; the whole score row of a tile lives in registers, i.e. a plain (not online) softmax.
//   * S^T = K . Q^T with v_mfma_f32_16x16x32_bf16 (keys on the accumulator rows, queries on the lanes), so
//     every lane owns ONE query: the softmax reductions are per-lane loops plus two cross-lane steps.
//   * In that orientation the accumulator registers are already the B operand of the second product
//     O^T = V^T . P^T (k order permuted: slot j of lane group g is key 32s + 16(j>>2) + 4g + (j&3)), so P never
//     leaves registers.  V is read from its row-major LDS image with ds_read_b64_tr_b16 (hardware transpose),
//     addressed with the same key permutation.
//   * LDS images are XOR-swizzled by 16-byte chunk: K for ds_read_b128 fragment reads, V for the transposed reads.
// Policy kernel (attn_policy): head_dim 96, at most 32 keys; one wave per (episode, head, query), fp32.
#include <type_traits>

#include "common.hpp"
#include "otter_hip.h"

namespace {

struct AttnArgs {
    const void* qkv; long ld;        // rows (seq*L + t), q|k|v blocks of heads*64 each (bf16 or fp16)
    void* out; long ldo;
    int L, heads, causal;
    float scale;
};

__device__ __forceinline__ int k_swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
__device__ __forceinline__ int v_swz(int row, int chunk) { return chunk ^ (((row >> 1) & 3) << 1); }

// NT = number of 16-key tiles computed (also the number of 16-query tiles); LS = the sequence length when it is
// known at compile time (0: taken from the arguments).  With LS the validity of every key is a compile-time fact
// except in the last, partial tile, so the softmax carries no masking work for the full tiles.
//
// The three phases of one 16-query tile, shared by the two kernels below.  Lane (g, li) owns query column li.

// S^T = K . Q^T for all NT key tiles.  A wave issues in order and has at most one partner wave on its SIMD, so LDS
// latency is hidden by hand: the K fragments of the next batch of key tiles are requested before the MFMAs of the
// current batch are issued.  Every key tile is computed, also rows past L: a wave-uniform skip would put a branch
// between an MFMA and the first read of its accumulator.
// NQ query tiles (1 or 2) share every K fragment read: the LDS serves 8 waves per CU, and a query tile that reads all of K
// and all of V for itself (68 KiB per tile and wave) costs as much LDS time per round of tiles as its MFMAs or its
// softmax take; with two tiles per pass that is halved.
template <typename T, int NT, int NQ>
__device__ __forceinline__ void tile_scores(const char* sK, int k_row, int k_sw, int g, const typename Half16<T>::v8 (&qf)[2][2], f32x4 (&s)[NQ][NT]) {
    typedef typename Half16<T>::v8 v8;
    constexpr int KB = NQ == 2 ? 2 : 4, NB = (NT + KB - 1) / KB;
    v8 kf[2][KB][2];
    auto read_k = [&](int b, v8 (&dst)[KB][2]) {
#pragma unroll
        for (int i = 0; i < KB; ++i)
            if (b * KB + i < NT) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    dst[i][ks] = *(const v8*)(sK + (b * KB + i) * 2048 + k_row + (((ks * 4 + g) ^ k_sw) << 4));
            }
    };
    read_k(0, kf[0]);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (b + 1 < NB) read_k(b + 1, kf[(b + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < KB; ++i)
            if (b * KB + i < NT) {
                const int kt = b * KB + i;
#pragma unroll
                for (int t = 0; t < NQ; ++t) s[t][kt] = Half16<T>::mfma32(kf[b & 1][i][0], qf[t][0], (f32x4){0.f, 0.f, 0.f, 0.f});
            }
#pragma unroll
        for (int i = 0; i < KB; ++i)
            if (b * KB + i < NT) {
                const int kt = b * KB + i;
#pragma unroll
                for (int t = 0; t < NQ; ++t) s[t][kt] = Half16<T>::mfma32(kf[b & 1][i][1], qf[t][1], s[t][kt]);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// mask + exponentials over keys (per lane: one query); key kt*16 + 4g + r is valid iff kt*16 + r < lim.  The row sums come
// from the matrix pipe in tile_pv (an all-ones A operand against the same P fragments that multiply V: nine MFMAs instead
// of 68 VALU additions, in a loop whose pace the vector issue port sets), so the normaliser is the sum of exactly the
// rounded probabilities that are multiplied into V.
template <int NT, int LS>
__device__ __forceinline__ void tile_softmax(f32x4 (&s)[NT], int lim, float c2) {
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool known_valid = LS && (kt * 16 + 15 < LS);      // every lane's key of this tile is < L
            if (!known_valid) s[kt][r] = (kt * 16 + r < lim) ? s[kt][r] : -INFINITY;
            mx = fmaxf(mx, s[kt][r]);
        }
    mx = rows_max(mx);
    const float off = -mx * c2;                  // softmax(scale*s) = 2^(c2*s - c2*max) / sum
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[kt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], c2, off));   // 2^(-inf) = 0 for masked keys
}

// O^T = V^T . P^T and the row sums for NQ query tiles that share the V^T fragment reads; inv[t] = 1 / sum of this lane's
// query of tile t.  V^T fragments (two transposed reads per MFMA) are requested a key-tile pair ahead of their MFMAs.
template <typename T, int NT, int NQ>
__device__ __forceinline__ void tile_pv(const char* sV, const f32x4 (&s)[NQ][NT], int g, int li, f32x4 (&o)[NQ][4], float (&inv)[NQ]) {
    typedef typename Half16<T>::v8 v8;
    constexpr int NP = (NT + 1) / 2;               // key-tile pairs of the second product (K = 32 per MFMA)
    const int vr = li >> 2;                        // lane li of group g addresses row r0 + (li>>2), bytes (li&3)*8 of the d-tile
    f32x4 osum[NQ];
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        osum[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    v8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (T)1.0f;
    v8 vf[2][4];
    auto read_v = [&](int s2, v8 (&dst)[4]) {
        const bool has_hi = 2 * s2 + 1 < NT;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int rlo = 32 * s2 + 4 * g + vr, rhi = rlo + 16;
            const int cl = dt * 2 + ((li & 3) >> 1), sub = (li & 1) * 8;
            union { struct { s16x4 a, b; } h; v8 v; } u;
            u.h.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(sV + rlo * 128 + (v_swz(rlo, cl) << 4) + sub));
            u.h.b = (s16x4){0, 0, 0, 0};
            if (has_hi)
                u.h.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(sV + rhi * 128 + (v_swz(rhi, cl) << 4) + sub));
            dst[dt] = u.v;
        }
    };
    read_v(0, vf[0]);
#pragma unroll
    for (int s2 = 0; s2 < NP; ++s2) {
        constexpr T zero = (T)0.0f;
        const bool has_hi = 2 * s2 + 1 < NT;
        if (s2 + 1 < NP) read_v(s2 + 1, vf[(s2 + 1) & 1]);
        v8 pf[NQ];
#pragma unroll
        for (int t = 0; t < NQ; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[t][r] = Half16<T>::from_unit(s[t][2 * s2][r]);
                pf[t][4 + r] = has_hi ? Half16<T>::from_unit(s[t][has_hi ? 2 * s2 + 1 : 0][r]) : zero;
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int t = 0; t < NQ; ++t) o[t][dt] = Half16<T>::mfma32(vf[s2 & 1][dt], pf[t], o[t][dt]);
#pragma unroll
        for (int t = 0; t < NQ; ++t) osum[t] = Half16<T>::mfma32(ones, pf[t], osum[t]);   // every accumulator row = sum over this pair's 32 keys, per query (lane)
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NQ; ++t) inv[t] = __builtin_amdgcn_rcpf(osum[t][0]);
}

// (a convex combination of 16-bit values: no saturation needed on the way back to 16 bits)
// Lane (g, li) holds columns dt*16 + 4g .. + 3 of its query's row for the four d-tiles: 8-byte pieces, four lanes and four
// store instructions per 128-byte row, every instruction writing 32-byte fragments of 16 different lines.  One
// v_permlane16_swap per packed register pairs the pieces of neighbouring lane groups instead: lane groups 0 / 2 end up with
// 8 consecutive columns of an even d-tile, groups 1 / 3 with 8 consecutive columns of the odd d-tile next to it, so a row
// leaves in two instructions of 64 contiguous bytes (16 bytes per lane).  `valid`: the lane's query exists (the exchange
// itself needs every lane).
template <typename T>
__device__ __forceinline__ void tile_store(T* row_out, const f32x4 (&o)[4], float inv, int g, bool valid) {
    typedef typename Half16<T>::v8 v8;
    union { typename Half16<T>::v4 h; unsigned w[2]; } pk[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) pk[dt].h[r] = (T)(o[dt][r] * inv);
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        union { unsigned w[4]; v8 v; } out;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            // rows of 16 lanes = lane groups: group 2k+1 of the first operand trades places with group 2k of the second
            const auto sw = __builtin_amdgcn_permlane16_swap(pk[2 * pr].w[w], pk[2 * pr + 1].w[w], false, false);
            out.w[w] = sw[0];            // even group: its own even-tile piece      odd group: the even group's odd-tile piece
            out.w[2 + w] = sw[1];        // even group: the odd group's even-tile piece   odd group: its own odd-tile piece
        }
        const int col = (g & 1) ? (2 * pr + 1) * 16 + 4 * (g - 1) : 2 * pr * 16 + 4 * g;
        if (valid) *(v8*)(row_out + col) = out.v;
    }
}

// One workgroup per (sequence, head): any length up to 592, optional causal mask.  A wave works on its query tiles two at a
// time (NT <= 18: the score rows of two tiles fit the registers), a last odd one alone.
template <typename T, int NT, int LS, int NW>      // NW waves per workgroup share the query tiles
__global__ __launch_bounds__(NW * 64, NT <= 18 ? 2 : 1) void attn_d64(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Half16<T>::v8 v8;
    const T* qkv = (const T*)p.qkv;
    T* outp = (T*)p.out;
    constexpr int LK = NT * 16;
    constexpr bool DUAL = NT <= 18;
    char* sK = smem;
    char* sV = smem + LK * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seq = blockIdx.x / p.heads, h = blockIdx.x % p.heads;
    const int L = LS ? LS : p.L;
    const bool causal = LS ? false : (p.causal != 0);
    const long row0 = (long)seq * L;
    const int hd = p.heads * 64;
    const int g = lane >> 4, li = lane & 15;

    // first query tiles of this wave: issue their loads before the K/V staging so both latencies overlap
    auto load_q = [&](int qt, v8 (&qf)[2]) {
        const int qc = min(qt * 16 + li, L - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[ks] = *(const v8*)(qkv + (row0 + qc) * p.ld + h * 64 + ks * 32 + g * 8);
    };
    v8 qf[2][2];
    // NT is rarely a multiple of NW, so some wave gets one tile more; which one rotates with the workgroup id so that the
    // workgroups sharing a CU do not all put their extra tile on the same SIMD
    const int wslot = (wave + (blockIdx.x ^ (blockIdx.x >> 3) ^ (blockIdx.x >> 8))) % NW;
    load_q(wslot < NT ? wslot : 0, qf[0]);
    if (DUAL) load_q(wslot + NW < NT ? wslot + NW : 0, qf[1]);

    // ---- stage K and V of this head by LDS-DMA: one instruction = 8 rows x 128 bytes, lane-linear destination, the
    //      chunk swizzle applied on the source address.  Rows >= L repeat row L-1 (finite data; their keys are masked). ----
    for (int grp = wave; grp < LK / 8; grp += NW) {
        const int r = grp * 8 + (lane >> 3), pc = lane & 7;
        const T* base = qkv + (row0 + min(r, L - 1)) * p.ld + h * 64;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(base + hd + (k_swz(r, pc) << 3)), LDS_PTR(sK + grp * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(base + 2 * hd + (v_swz(r, pc) << 3)), LDS_PTR(sV + grp * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // K fragment read: row kt*16 + li, logical chunk ks*4 + g
    const int k_row_off = li * 128;
    const int k_sw = (li >> 1) & 7;
    const float c2 = p.scale * 1.44269504088896341f;

    // NQ tiles qt, qt + NW (the second only with NQ = 2); nxt = this wave's tile after them (its query rows are requested
    // while the softmax runs)
    auto process = [&](auto nq_tag, int qt, int nxt) {
        constexpr int NQ = decltype(nq_tag)::value;
        // The K fragments do not depend on the query tile; left alone the compiler keeps all of them in registers
        // across the loop, which leaves too few to keep the V^T reads of the second product in
        // flight.  An opaque copy of the row offset keeps the (cheap) LDS reads inside the loop.
        int k_row = k_row_off;
        asm volatile("" : "+v"(k_row));
        f32x4 s[NQ][NT];
        tile_scores<T, NT, NQ>(sK, k_row, k_sw, g, qf, s);
        if (nxt < NT) load_q(nxt, qf[0]);           // lands while the softmax runs
        if (DUAL && nxt + NW < NT) load_q(nxt + NW, qf[1]);
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
            const int q = (qt + t * NW) * 16 + li;          // this lane's query
            tile_softmax<NT, LS>(s[t], (causal ? min(L, q + 1) : L) - 4 * g, c2);
        }
        f32x4 o[NQ][4];
        float inv[NQ];
        tile_pv<T, NT, NQ>(sV, s, g, li, o, inv);
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
            const int q = (qt + t * NW) * 16 + li;
            tile_store<T>(outp + (row0 + min(q, L - 1)) * p.ldo + h * 64, o[t], inv[t], g, q < L);
        }
    };
    int qt = wslot;
    if constexpr (DUAL) {
        for (; qt + NW < NT; qt += 2 * NW) process(std::integral_constant<int, 2>{}, qt, qt + 2 * NW);
    }
    for (; qt < NT; qt += NW) process(std::integral_constant<int, 1>{}, qt, qt + NW);
}

template <typename T, int NT, int LS, int NW = 4>
int launch_d64(const AttnArgs& a, int n_seq, hipStream_t s) {
    constexpr int lds = NT * 16 * 128 * 2;
    auto kern = attn_d64<T, NT, LS, NW>;
    if (int rc = otter_reserve_lds((const void*)kern, lds, "otter_attn_fwd_d64")) return rc;
    hipLaunchKernelGGL(kern, dim3(n_seq * a.heads), dim3(NW * 64), lds, s, a);
    OTTER_CHECK_LAUNCH("otter_attn_fwd_d64");
    return OTTER_OK;
}

// ------------------------------------------------------------------------------------------------
// Policy attention: fp32 q|k|v rows of width 3*dim (dim = heads*hd), causal, window of at most 64 keys.
// One wave per (episode b, head h, query i).  Query i of the call sits at window position pos0 + i and
// attends keys 0 .. pos0+i of the same episode's rows (the rows double as the GPU-resident K/V history).
struct PolicyAttnArgs {
    const float* qkv; long ld;      // row (b*S + t)
    float* out; long ldo;           // row (b*Tq + i)
    int S, Tq, pos0, heads, hd;
    float scale;
};

// q . k over hd floats (hd % 4 == 0) with eight 16-byte loads of each row in flight at a time: a loop with a run-time bound
// that loads, waits and multiplies once per 16 bytes spends one memory round trip per iteration (24 of them at hd = 96).
__device__ __forceinline__ float dot_rows(const float* a, const float* b, int hd) {
    float acc = 0.f;
    int d = 0;
    for (; d + 32 <= hd; d += 32) {
        f32x4 av[8], bv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { av[i] = *(const f32x4*)(a + d + 4 * i); bv[i] = *(const f32x4*)(b + d + 4 * i); }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += av[i][0] * bv[i][0] + av[i][1] * bv[i][1] + av[i][2] * bv[i][2] + av[i][3] * bv[i][3];
    }
    for (; d < hd; d += 4) {
        const f32x4 av = *(const f32x4*)(a + d), bv = *(const f32x4*)(b + d);
        acc += av[0] * bv[0] + av[1] * bv[1] + av[2] * bv[2] + av[3] * bv[3];
    }
    return acc;
}

__global__ __launch_bounds__(256) void attn_policy(PolicyAttnArgs p, int total) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= total) return;
    const int i = w % p.Tq, h = (w / p.Tq) % p.heads, b = w / (p.Tq * p.heads);
    const int dim = p.heads * p.hd;
    const int qpos = p.pos0 + i, nkeys = qpos + 1;
    const float* qrow = p.qkv + ((long)b * p.S + qpos) * p.ld + h * p.hd;
    float sc = -INFINITY;
    if (lane < nkeys) {
        const float* krow = p.qkv + ((long)b * p.S + lane) * p.ld + dim + h * p.hd;
        sc = dot_rows(qrow, krow, p.hd) * p.scale;
    }
    const float mx = wave_max(sc);
    const float e = (lane < nkeys) ? __expf(sc - mx) : 0.f;
    const float inv = 1.0f / wave_sum(e);
    // four value rows in flight per pass (same summation order as one at a time)
    const float* vbase = p.qkv + (long)b * p.S * p.ld + 2 * dim + h * p.hd;
    for (int d = lane; d < p.hd; d += 64) {
        float acc = 0.f;
        int j = 0;
        for (; j + 4 <= nkeys; j += 4) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = vbase[(long)(j + u) * p.ld + d];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += __shfl(e, j + u, 64) * v[u];
        }
        for (; j < nkeys; ++j) acc += __shfl(e, j, 64) * vbase[(long)j * p.ld + d];
        p.out[((long)b * p.Tq + i) * p.ldo + h * p.hd + d] = acc * inv;
    }
}

// Ragged cached step: row i is the newest position of episode idx[i] (window position lens[idx[i]] - 1).  Its fresh q|k|v
// row comes from a compact buffer; the wave stores it into the episode's cache row and attends over the cached rows
// before it plus the fresh one.  A wave reads only cache rows of earlier positions and writes only its own head's slice
// of the new row, so the store and the reads of one launch never meet.
struct PolicyRowsArgs {
    const float* fresh;    // [n_rows][ldf]: q | k | v of the new positions
    long ldf;
    float* cache;          // [n_episodes][S][ld]
    long ld;
    float* out;            // [n_rows][ldo]
    long ldo;
    const int* idx;
    const int* lens;
    int n_episodes, S, heads, hd;
    float scale;
};

__global__ __launch_bounds__(256) void attn_policy_rows(PolicyRowsArgs p, int total) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= total) return;
    const int h = w % p.heads, i = w / p.heads;
    const int dim = p.heads * p.hd;
    const int e = min(max(p.idx[i], 0), p.n_episodes - 1);
    const int pos = min(max(p.lens[e] - 1, 0), p.S - 1), nkeys = pos + 1;
    const float* qrow = p.fresh + (long)i * p.ldf + h * p.hd;
    const float* knew = qrow + dim;
    const float* vnew = qrow + 2 * dim;
    const float* hist = p.cache + (long)e * p.S * p.ld;
    float* crow = p.cache + ((long)e * p.S + pos) * p.ld + h * p.hd;
    for (int d = lane; d < p.hd; d += 64) {
        crow[d] = qrow[d];
        crow[dim + d] = knew[d];
        crow[2 * dim + d] = vnew[d];
    }
    float sc = -INFINITY;
    if (lane < nkeys) {
        const float* krow = (lane == pos) ? knew : hist + (long)lane * p.ld + dim + h * p.hd;
        sc = dot_rows(qrow, krow, p.hd) * p.scale;
    }
    const float mx = wave_max(sc);
    const float ex = (lane < nkeys) ? __expf(sc - mx) : 0.f;
    const float inv = 1.0f / wave_sum(ex);
    const float* vhist = hist + 2 * dim + h * p.hd;
    for (int d = lane; d < p.hd; d += 64) {
        float acc = 0.f;
        int j = 0;
        for (; j + 4 <= nkeys; j += 4) {                       // four value rows in flight per pass
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = (j + u == pos) ? vnew[d] : vhist[(long)(j + u) * p.ld + d];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += __shfl(ex, j + u, 64) * v[u];
        }
        for (; j < nkeys; ++j) acc += __shfl(ex, j, 64) * (j == pos ? vnew[d] : vhist[(long)j * p.ld + d]);
        p.out[(long)i * p.ldo + h * p.hd + d] = acc * inv;
    }
}

}  // namespace

template <typename T>
static int dispatch_d64(const AttnArgs& a, int n_seq, int L, hipStream_t s) {
    if (L == 257 && !a.causal) return launch_d64<T, 17, 257, 4>(a, n_seq, s);    // 4 waves measured faster than 5 or 6 (79 vs 88 / 82 us per layer)      // ViT-L/14 and ViT-B/... at 224: 256 patches + CLS
    if (L <= 32) return launch_d64<T, 2, 0>(a, n_seq, s);
    if (L <= 96) return launch_d64<T, 6, 0>(a, n_seq, s);
    if (L <= 288) return launch_d64<T, 18, 0>(a, n_seq, s);
    return launch_d64<T, 37, 0>(a, n_seq, s);      // ViT-L/14@336px: 577 tokens, K and V fill 148 of the 160 KiB: one workgroup per CU
}

extern "C" int otter_attn_fwd_d64(int dtype, const void* qkv, long ld, void* out, long ldo, int n_seq, int L, int heads,
                                  int causal, void* stream) {
    OTTER_REQUIRE(dtype == DT_BF16 || dtype == DT_F16, OTTER_ERR_INVALID_ARGUMENT, "otter_attn_fwd_d64: dtype %d (bf16 or fp16)", dtype);
    OTTER_REQUIRE(qkv && out, OTTER_ERR_INVALID_ARGUMENT, "otter_attn_fwd_d64: null operand");
    if (n_seq == 0) return OTTER_OK;
    OTTER_REQUIRE(n_seq > 0 && L > 0 && heads > 0, OTTER_ERR_INVALID_ARGUMENT, "otter_attn_fwd_d64: n_seq=%d L=%d heads=%d", n_seq, L, heads);
    OTTER_REQUIRE(L <= 592, OTTER_ERR_UNSUPPORTED_SHAPE, "otter_attn_fwd_d64: L=%d exceeds 592 keys", L);
    OTTER_REQUIRE(ld >= 3 * heads * 64 && ld % 8 == 0 && ldo >= heads * 64 && ldo % 4 == 0, OTTER_ERR_ALIGNMENT,
                  "otter_attn_fwd_d64: ld=%ld / ldo=%ld too small or misaligned for %d heads", ld, ldo, heads);
    OTTER_REQUIRE(aligned16(qkv) && aligned16(out), OTTER_ERR_ALIGNMENT, "otter_attn_fwd_d64: operands must be 16-byte aligned");
    OTTER_REQUIRE((long)n_seq * heads < (1L << 31), OTTER_ERR_UNSUPPORTED_SHAPE, "otter_attn_fwd_d64: grid too large");
    AttnArgs a{qkv, ld, out, ldo, L, heads, causal ? 1 : 0, 0.125f};
    hipStream_t s = (hipStream_t)stream;
    return dtype == DT_BF16 ? dispatch_d64<__bf16>(a, n_seq, L, s) : dispatch_d64<_Float16>(a, n_seq, L, s);
}

extern "C" int otter_attn_policy(const float* qkv, long ld, float* out, long ldo, int batch, int window, int n_query,
                                 int pos0, int heads, int head_dim, void* stream) {
    OTTER_REQUIRE(qkv && out, OTTER_ERR_INVALID_ARGUMENT, "otter_attn_policy: null operand");
    if (batch == 0 || n_query == 0) return OTTER_OK;
    OTTER_REQUIRE(batch > 0 && n_query > 0 && pos0 >= 0 && heads > 0 && head_dim > 0, OTTER_ERR_INVALID_ARGUMENT,
                  "otter_attn_policy: bad sizes");
    OTTER_REQUIRE(pos0 + n_query <= window && window <= 64, OTTER_ERR_UNSUPPORTED_SHAPE,
                  "otter_attn_policy: pos0+n_query=%d must fit the window=%d (<= 64)", pos0 + n_query, window);
    OTTER_REQUIRE(head_dim % 4 == 0 && ld >= 3L * heads * head_dim && ld % 4 == 0 && ldo >= (long)heads * head_dim,
                  OTTER_ERR_ALIGNMENT, "otter_attn_policy: head_dim must be a multiple of 4 and ld/ldo large enough");
    OTTER_REQUIRE(aligned16(qkv), OTTER_ERR_ALIGNMENT, "otter_attn_policy: qkv must be 16-byte aligned");
    PolicyAttnArgs a{qkv, ld, out, ldo, window, n_query, pos0, heads, head_dim, 1.0f / sqrtf((float)head_dim)};
    const long total = (long)batch * heads * n_query;
    OTTER_REQUIRE(total < (1L << 31), OTTER_ERR_UNSUPPORTED_SHAPE, "otter_attn_policy: grid too large");
    hipLaunchKernelGGL(attn_policy, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, (int)total);
    OTTER_CHECK_LAUNCH("otter_attn_policy");
    return OTTER_OK;
}

extern "C" int otter_attn_policy_rows(const float* fresh, long ldf, float* cache, long ld, float* out, long ldo, const int* idx,
                                      const int* lens, int n_rows, int n_episodes, int window, int heads, int head_dim, void* stream) {
    OTTER_REQUIRE(fresh && cache && out && idx && lens, OTTER_ERR_INVALID_ARGUMENT, "otter_attn_policy_rows: null operand");
    if (n_rows == 0) return OTTER_OK;
    OTTER_REQUIRE(n_rows > 0 && n_episodes > 0 && heads > 0 && head_dim > 0 && window > 0 && window <= 64, OTTER_ERR_UNSUPPORTED_SHAPE,
                  "otter_attn_policy_rows: bad sizes (window=%d, at most 64)", window);
    OTTER_REQUIRE(head_dim % 4 == 0 && ld >= 3L * heads * head_dim && ld % 4 == 0 && ldf >= 3L * heads * head_dim && ldf % 4 == 0 &&
                  ldo >= (long)heads * head_dim, OTTER_ERR_ALIGNMENT,
                  "otter_attn_policy_rows: head_dim must be a multiple of 4 and the leading dimensions large enough");
    OTTER_REQUIRE(aligned16(fresh) && aligned16(cache), OTTER_ERR_ALIGNMENT, "otter_attn_policy_rows: buffers must be 16-byte aligned");
    OTTER_REQUIRE(fresh != cache, OTTER_ERR_INVALID_ARGUMENT, "otter_attn_policy_rows: the fresh rows and the cache must be different buffers");
    PolicyRowsArgs a{fresh, ldf, cache, ld, out, ldo, idx, lens, n_episodes, window, heads, head_dim, 1.0f / sqrtf((float)head_dim)};
    const long total = (long)n_rows * heads;
    OTTER_REQUIRE(total < (1L << 31), OTTER_ERR_UNSUPPORTED_SHAPE, "otter_attn_policy_rows: grid too large");
    hipLaunchKernelGGL(attn_policy_rows, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, (int)total);
    OTTER_CHECK_LAUNCH("otter_attn_policy_rows");
    return OTTER_OK;
}
