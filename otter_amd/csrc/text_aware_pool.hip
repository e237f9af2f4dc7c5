// Fused text-aware visual extraction (memory-bound; one pass over the patch tokens).
//
// Per unit (observation m, camera c):   patches X (P x D, bf16 or fp16, un-normalised), text T (K<=16 rows x D, fp32)
//   S    = T^ . X^^T              cosine similarity (T^, X^ = L2-normalised rows)
//   A    = softmax_over_patches(S / clamp(temperature, 0, 100))
//   out  = A . (X^ + pos_emb)     (K x D, fp32)
//
// One workgroup (4 waves) per unit; LDS and registers are budgeted so that two workgroups share a CU and overlap each
// other's dependent chains.  The patch tokens stream HBM -> LDS exactly once, in 16-patch tiles by LDS-DMA
// (global_load_lds_dwordx4) through two buffers, so one tile is in flight behind the one being consumed and the only
// wait per tile is an s_waitcnt.  Per tile:
//   * S^T tile (patch x text) and the Gram diagonal |x|^2 come from v_mfma_f32_16x16x32 over D; each wave
//     reduces its quarter of D (its text fragments stay in registers for the whole unit) and the four partial
//     tiles are summed through LDS (identical softmax state in all waves afterwards);
//   * online softmax over patches (running max / sum per text token, one text token per lane), in log2 units;
//   * O^T += X^T . (p/|x|)  +  pos^T . p   with v_mfma_f32_16x16x16: the S^T accumulator layout
//     (patch 4g+r on the registers, text token on the lane) is exactly that instruction's B operand, X^T comes
//     from the row-major LDS tile by ds_read_b64_tr_b16, and pos^T comes from a 16-bit copy of pos_emb packed in
//     fragment order (otter_pos_emb_pack), so each lane fetches two d-tiles with one coalesced 16-byte load from
//     L2 (issued at the top of the tile, used at its end).  Each wave owns D/4 output columns.
// LDS rows are D*2 bytes (a multiple of 256), so the 16-byte chunk index is XORed with f(row) (below) which
// makes both the ds_read_b128 fragment reads and the transposed reads bank-conflict free.  The LDS image is
// lane-linear (LDS-DMA requirement); the swizzle is applied to the per-lane source address.
#include <type_traits>

#include "common.hpp"
#include "otter_hip.h"

#ifndef TAVE_TILE_CPOL
#define TAVE_TILE_CPOL 2      // patch tiles are read once: nt (streaming) policy, +2 % at chip-filling sizes
#endif
namespace {

struct TaveArgs {
    const void* patches;        // unit u at patches + u*unit_stride, [P][D] row-major, bf16 or fp16
    long unit_stride;
    const void* text;           // [n_text][K][D], fp32 or (TXT16) the patches' 16-bit type
    const void* posF;           // [cams][P/16][4 waves][NDT/2][64 lanes][8]: pos_emb in MFMA-fragment order (otter_pos_emb_pack)
    const float* temperature;   // [cams]
    void* out;                  // [cams][M][K][D], fp32 or (OUT16) the patches' 16-bit type
    int units, cams, P, D, K, t_rep, M;
};

// chunk swizzle: bijection on 4 bits with f(r)>>1 injective on each aligned octet of rows and
// f({0..3,12..15}) closed under ^1  (see header comment)
__device__ __forceinline__ int f_swz(int r) {
    const int lo = r & 7, hi = (r >> 3) & 1;
    return (((lo + 4 * hi) & 7) << 1) | hi;
}

// NW waves per workgroup: each reduces D/NW columns of the similarity product and owns D/NW output columns (NDTW d-tiles
// of 16).  NST = patch-tile buffers in LDS (prefetch distance NST-1 tiles).  WGS = workgroups the register budget lets
// share a CU: the waves of co-resident workgroups overlap each other's dependent chains.
// TXT16 / OUT16: the text tokens arrive / the result leaves in the patches' 16-bit type instead of fp32 (a unit then moves
// 439,296 bytes instead of 485,376 at P = 256, K = 15, D = 768).
template <typename T, int D, int NW, int NST, int WGS, bool TXT16, bool OUT16>
__global__ __launch_bounds__(NW * 64, WGS) void tave_kernel(TaveArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Half16<T>::v8 v8;
    typedef typename Half16<T>::v4 v4;
    constexpr int NDTW = D / 16 / NW;           // d-tiles per wave
    constexpr int ROWB = D * 2;                 // bytes per LDS row
    constexpr int TILE_B = 16 * ROWB;           // one 16-row image
    constexpr int CPR = D / 8;                  // 16-byte chunks per row
    constexpr int NLD = TILE_B / (NW * 64 * 16);// LDS-DMA instructions per thread per tile
    constexpr int PF = NST - 1;                 // prefetch distance in tiles
    constexpr int KSW = D / 32 / NW;            // K-steps (32 columns) of the similarity product per wave
    constexpr int TRW = 16 / NW;                // text rows normalised per wave
    static_assert(NDTW >= 2 && NDTW % 2 == 0 && KSW >= 1 && NLD >= 1 && NST >= 2, "unsupported D / wave count");
    char* sX = smem;
    char* sT = smem + (NST - 1) * TILE_B;       // text image: borrows the buffer that the first prefetch fills last
    float* sP = (float*)(smem + NST * TILE_B);  // [NW waves][64 lanes][4] partial S^T tiles, then [NW waves][16] partial |x|^2

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int unit = blockIdx.x;
    const int m = unit / p.cams, cam = unit % p.cams;
    const int P = p.P, nt = (P + 15) >> 4;     // the last tile may be partial: its missing rows are masked out of the softmax
    const T* X = (const T*)p.patches + (long)unit * p.unit_stride;
    const T* posF = (const T*)p.posF + (long)cam * (nt * 16) * D;
    const int frag_row = li * ROWB;
    const int fsw = f_swz(li);

    // ---- per-lane LDS-DMA source offsets (bytes within a tile), identical for every tile ----
    int src_off[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int q = (wave * NLD + i) * 64 + lane;          // physical chunk index in the tile
        const int r = q / CPR, pc = q % CPR;
        src_off[i] = r * ROWB + ((pc ^ f_swz(r)) << 4);
    }
    auto stage = [&](int buf, int tile) {
        const char* base = (const char*)(X + (long)tile * 16 * D);
        const int lim = (P - tile * 16) * ROWB - 16;         // chunks past the unit's last row re-read its last chunk (finite, masked)
        char* dst = sX + buf * TILE_B;
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(base + min(src_off[i], lim)), LDS_PTR(dst + (wave * NLD + i) * 1024), 16, 0, TAVE_TILE_CPOL);
    };
    stage(0, 0);
    if (PF > 1 && nt > 1) stage(1, 1);

    // ---- text rows: L2-normalise in fp32, store as 16-bit (swizzled) ; rows >= K are zero ----
    {
        const int b = m / p.t_rep;
#pragma unroll
        for (int rr = 0; rr < TRW; ++rr) {               // unrolled: the loads of the wave's rows are in flight together
            const int r = wave * TRW + rr;
            const long row_off = ((long)b * p.K + r) * D;
            float sq = 0.f;
            f32x4 v[(D / 4 + 63) / 64];
#pragma unroll
            for (int i = 0; i < (D / 4 + 63) / 64; ++i) {
                const int idx = i * 64 + lane;
                v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (idx < D / 4 && r < p.K) {
                    if constexpr (TXT16) {
                        const v4 h = *(const v4*)((const T*)p.text + row_off + idx * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[i][e] = (float)h[e];
                    } else {
                        v[i] = *(const f32x4*)((const float*)p.text + row_off + idx * 4);
                    }
                }
                sq += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
            }
            sq = wave_sum(sq);
            const float rn = (r < p.K) ? 1.0f / sqrtf(sq) : 0.f;
#pragma unroll
            for (int i = 0; i < (D / 4 + 63) / 64; ++i) {
                const int idx = i * 64 + lane;           // 4-element group index
                if (idx < D / 4) {
                    v4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = Half16<T>::from(v[i][e] * rn);
                    const int chunk = idx >> 1;
                    *(v4*)(sT + r * ROWB + ((chunk ^ f_swz(r)) << 4) + (idx & 1) * 8) = o;
                }
            }
        }
    }
    // each wave keeps the text fragments of its D/NW columns in registers for the whole unit
    __syncthreads();
    v8 tfr[KSW];
#pragma unroll
    for (int kk = 0; kk < KSW; ++kk) tfr[kk] = *(const v8*)(sT + frag_row + ((((wave * KSW + kk) * 4 + g) ^ fsw) << 4));
    const float temp = fminf(fmaxf(p.temperature[cam], 0.f), 100.f);
    const float inv_temp = 1.4426950408889634f / temp;      // logits are kept in log2 units: softmax through v_exp_f32 (2^x)

    f32x4 o[NDTW];
#pragma unroll
    for (int dt = 0; dt < NDTW; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float run_max = -INFINITY, run_sum = 0.f;

    // transposed-read addressing: lane li of group g -> row 4g + (li>>2), 8 bytes at column 4*(li&3) of the d-tile
    const int tr_row = 4 * g + (li >> 2);
    const int tr_fsw = f_swz(tr_row);
    const int tr_sub = (li & 1) * 8;

    // One tile.  PREFETCH (compile-time): tile t+2 exists and is fetched; keeping it out of a runtime branch lets the
    // compiler count the loads in flight, so its wait for the pos^T loads does not drain the prefetch.
    auto tile_step = [&](int t, auto prefetch_tag, auto more_tag) {
        constexpr bool PREFETCH = decltype(prefetch_tag)::value, MORE = decltype(more_tag)::value;
        // Tile t has landed once all but the youngest tile's loads are done (vector-memory operations retire in
        // issue order; the pos^T loads of the previous iteration were consumed there).  The barrier also tells
        // that every wave is done with tile t-1, whose buffer tile t+2 is about to overwrite.
        if constexpr (MORE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");     // tile t+1 may still be in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const char* xb = sX + (t % NST) * TILE_B;

        // pos_emb of this tile for the wave's columns, pre-packed in fragment order: one coalesced 16-byte load per lane
        // covers two d-tiles (L2-resident table).  Issued first, consumed after the similarity phase; from inline asm
        // because the compiler would wait for them with vmcnt(0), draining the tile prefetch issued right after --
        // the counted wait before their first use is placed by hand below.
        s16x8 pt[NDTW / 2];
#pragma unroll
        for (int dp = 0; dp < NDTW / 2; ++dp) {
            const T* src = posF + (((long)t * NW + wave) * (NDTW / 2) + dp) * 512 + lane * 8;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pt[dp]) : "v"(src) : "memory");
        }
        if constexpr (PREFETCH) stage((t + PF) % NST, t + PF);

        // ---- partial S^T (patch x text) and Gram diagonal over this wave's quarter of D ----
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, gd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KSW; ++kk) {
            const int ks = wave * KSW + kk;
            const int coff = (((ks * 4 + g) ^ fsw) << 4);
            const v8 xf = *(const v8*)(xb + frag_row + coff);
            s = Half16<T>::mfma32(xf, tfr[kk], s);
            gd = Half16<T>::mfma32(xf, xf, gd);
        }
        float* pp = sP;                                  // reused every tile: the tile's first barrier separates the uses
        float* pg = sP + NW * 256;                       // [NW waves][16 patches]: the partial Gram diagonals
        *(f32x4*)(pp + (wave * 64 + lane) * 4) = s;
        // |x|^2 of patch 4g+r sits on lane (g, li = 4g+r), Gram register r
        if ((li >> 2) == g) pg[wave * 16 + li] = (li & 2) ? ((li & 1) ? gd[3] : gd[2]) : ((li & 1) ? gd[1] : gd[0]);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        s = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 n2 = s;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            s += *(const f32x4*)(pp + (w * 64 + lane) * 4);
            n2 += *(const f32x4*)(pg + w * 16 + 4 * g);
        }
        float a1[4], a2[4];
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float rn = __builtin_amdgcn_rsqf(fmaxf(n2[r], 1e-24f));  // F.normalize: x / max(|x|, 1e-12)
            a1[r] = rn;
            a2[r] = (t * 16 + 4 * g + r < P) ? s[r] * rn * inv_temp : -INFINITY;   // logit of (patch 4g+r, text li)
            tmax = fmaxf(tmax, a2[r]);
        }
        tmax = rows_max(tmax);
        const float new_max = fmaxf(run_max, tmax);
        const float alpha = __builtin_amdgcn_exp2f(run_max - new_max);     // first tile: exp(-inf) = 0
        run_max = new_max;
        float psum = 0.f;
        s16x4 b1, b2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pr = __builtin_amdgcn_exp2f(a2[r] - new_max);
            psum += pr;
            b1[r] = bits16(Half16<T>::from(pr * a1[r]));
            b2[r] = bits16(Half16<T>::from(pr));
        }
        run_sum = run_sum * alpha + psum;                   // per-lane partial (its own 4 patches per tile)
#pragma unroll
        for (int dt = 0; dt < NDTW; ++dt) o[dt] *= alpha;

        // ---- O^T += X^T (p/|x|) + pos^T p ----
        // the pos^T loads are older than the NLD prefetch instructions issued after them (in-order retirement);
        // the "+v" operands tie every later use of pt[] to this wait
#pragma unroll
        for (int dp = 0; dp < NDTW / 2; ++dp) {
            if constexpr (PREFETCH) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(pt[dp]) : "n"(NLD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(pt[dp]) : : "memory");
        }
        // X^T fragments by ds_read_b64_tr_b16, also from inline asm: after an LDS-DMA the compiler puts vmcnt(0) in
        // front of the builtin form of this read (it cannot tell that the buffers differ), draining the prefetch
        s16x4 xt[NDTW];
#pragma unroll
        for (int dt = 0; dt < NDTW; ++dt) {
            const int d0 = (wave * NDTW + dt) * 16;
            const int chunk = (d0 >> 3) + ((li & 3) >> 1);
            const unsigned addr = (unsigned)(uintptr_t)LDS_PTR(xb + tr_row * ROWB + ((chunk ^ tr_fsw) << 4) + tr_sub);
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(xt[dt]) : "v"(addr) : "memory");
        }
#pragma unroll
        for (int dt = 0; dt < NDTW; ++dt) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xt[dt]) : : "memory");
#pragma unroll
        for (int dt = 0; dt < NDTW; ++dt) {
            o[dt] = Half16<T>::mfma16(xt[dt], b1, o[dt]);
            const s16x8 pp2 = pt[dt >> 1];
            const s16x4 ph = (dt & 1) ? (s16x4){pp2[4], pp2[5], pp2[6], pp2[7]} : (s16x4){pp2[0], pp2[1], pp2[2], pp2[3]};
            o[dt] = Half16<T>::mfma16(ph, b2, o[dt]);
        }
    };
    {
        using Y = std::true_type;
        using N = std::false_type;
        int t = 0;
        if constexpr (NST == 3) {                  // two tiles ahead
            for (; t + 2 < nt; ++t) tile_step(t, Y{}, Y{});
            if (t + 1 < nt) { tile_step(t, N{}, Y{}); ++t; }
            if (t < nt) tile_step(t, N{}, N{});
        } else {                                   // one tile ahead: nothing younger than tile t is in flight when it is awaited
            for (; t + 1 < nt; ++t) tile_step(t, Y{}, N{});
            if (t < nt) tile_step(t, N{}, N{});
        }
    }

    run_sum = rows_sum(run_sum);
    const float inv = 1.0f / run_sum;
    if (li < p.K) {
        const long off = (((long)cam * p.M + m) * p.K + li) * D;
#pragma unroll
        for (int dt = 0; dt < NDTW; ++dt) {
            const int d0 = (wave * NDTW + dt) * 16;
            const f32x4 r = o[dt] * inv;
            if constexpr (OUT16) {
                v4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = Half16<T>::from(r[e]);
                *(v4*)((T*)p.out + off + d0 + 4 * g) = h;
            } else {
                *(f32x4*)((float*)p.out + off + d0 + 4 * g) = r;
            }
        }
    }
}

template <typename T, int D, int NW, int NST, int WGS, bool TXT16, bool OUT16>
int launch_tave_n(const TaveArgs& a, hipStream_t s) {
    constexpr int lds = NST * 16 * D * 2 + NW * (64 * 4 + 16) * 4;   // patch tiles (text image borrows one) + partial-tile exchange
    auto kern = tave_kernel<T, D, NW, NST, WGS, TXT16, OUT16>;
    if (int rc = otter_reserve_lds((const void*)kern, lds, "otter_text_aware_pool_fwd")) return rc;
    hipLaunchKernelGGL(kern, dim3(a.units), dim3(NW * 64), lds, s, a);
    OTTER_CHECK_LAUNCH("otter_text_aware_pool_fwd");
    return OTTER_OK;
}

template <typename T, int D>
int launch_tave(const TaveArgs& a, int io16, hipStream_t s) {
    // Two buffers of 16 patches, two workgroups per CU: measured against three buffers (same occupancy), against 8 waves
    // per workgroup and against a 168-register build that admits a third workgroup (24 registers spill at D = 768:
    // 273 us against 183 us at 2048 units), this is the fastest form at every unit count; the per-tile dependent chain,
    // not the prefetch depth or the byte count of the fp32 edges (16-bit edges: same time), sets the pace.
    switch (io16 & 3) {
        case 0: return launch_tave_n<T, D, 4, 2, 2, false, false>(a, s);
        case 1: return launch_tave_n<T, D, 4, 2, 2, true, false>(a, s);
        case 2: return launch_tave_n<T, D, 4, 2, 2, false, true>(a, s);
        default: return launch_tave_n<T, D, 4, 2, 2, true, true>(a, s);
    }
}

// pos_emb [P][D] fp32 -> 16-bit table in the order tave_kernel's lanes consume it (ndt = D/64):
//   packed[((t*4 + w)*(ndt/2) + dp)*64 + lane][h*4 + j] = pos[16t + 4g + j][(w*ndt + 2dp + h)*16 + li],  g = lane>>4, li = lane&15
// (rows past P are zero: P is rounded up to whole 16-patch tiles)
// i.e. 512-element blocks, one per (tile t, pair of d-tiles), pairs in column order -- independent of the wave count.
// (setup-time kernel, one thread per element)
template <typename T>
__global__ void pos_pack_kernel(const float* pos, T* packed, int P, int D, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ndt = D / 64;
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    long rest = i >> 9;
    const int dp = (int)(rest % (ndt / 2)); rest /= (ndt / 2);
    const int w = (int)(rest & 3), t = (int)(rest >> 2);
    const int h = e >> 2, j = e & 3, g = lane >> 4, li = lane & 15;
    const int row = 16 * t + 4 * g + j;
    packed[i] = Half16<T>::from(row < P ? pos[(long)row * D + (w * ndt + 2 * dp + h) * 16 + li] : 0.f);
}

template <typename T>
int dispatch_tave(const TaveArgs& a, int D, int io16, hipStream_t s) {
    switch (D) {
        case 128: return launch_tave<T, 128>(a, io16, s);
        case 512: return launch_tave<T, 512>(a, io16, s);
        case 768: return launch_tave<T, 768>(a, io16, s);
        case 1024: return launch_tave<T, 1024>(a, io16, s);
    }
    otter_set_error("otter_text_aware_pool_fwd: D=%d unsupported (128, 512, 768, 1024)", D);
    return OTTER_ERR_UNSUPPORTED_SHAPE;
}

}  // namespace

static int tave_entry(const char* who, int dtype, const void* patches, long unit_stride, const void* text, const void* pos_packed,
                      const float* temperature, void* out, int n_obs, int cams, int P, int D, int K, int t_rep, int io16, void* stream) {
    OTTER_REQUIRE(dtype == DT_BF16 || dtype == DT_F16, OTTER_ERR_INVALID_ARGUMENT, "%s: dtype %d (bf16 or fp16)", who, dtype);
    OTTER_REQUIRE(patches && text && pos_packed && temperature && out, OTTER_ERR_INVALID_ARGUMENT, "%s: null operand", who);
    OTTER_REQUIRE((io16 & ~3) == 0, OTTER_ERR_INVALID_ARGUMENT, "%s: io16=%d (bit 0: 16-bit text, bit 1: 16-bit result)", who, io16);
    if (n_obs == 0) return OTTER_OK;
    OTTER_REQUIRE(n_obs > 0 && cams > 0 && t_rep > 0, OTTER_ERR_INVALID_ARGUMENT, "%s: bad sizes", who);
    OTTER_REQUIRE(n_obs % t_rep == 0, OTTER_ERR_INVALID_ARGUMENT, "%s: n_obs=%d not a multiple of t_rep=%d", who, n_obs, t_rep);
    OTTER_REQUIRE(K >= 1 && K <= 16, OTTER_ERR_UNSUPPORTED_SHAPE, "%s: K=%d text tokens, supports 1..16", who, K);
    OTTER_REQUIRE(P >= 1 && (long)P * D < (1L << 30), OTTER_ERR_INVALID_ARGUMENT, "%s: P=%d patches", who, P);
    OTTER_REQUIRE(aligned16(patches) && aligned16(text) && aligned16(pos_packed) && aligned16(out), OTTER_ERR_ALIGNMENT,
                  "%s: operands must be 16-byte aligned", who);
    OTTER_REQUIRE((long)n_obs * cams < (1L << 31), OTTER_ERR_UNSUPPORTED_SHAPE, "%s: grid too large", who);
    OTTER_REQUIRE(unit_stride >= (long)P * D && unit_stride % 8 == 0, OTTER_ERR_ALIGNMENT,
                  "%s: unit_stride=%ld must be >= P*D and a multiple of 8 elements", who, unit_stride);
    TaveArgs a{patches, unit_stride, text, pos_packed, temperature, out, n_obs * cams, cams, P, D, K, t_rep, n_obs};
    hipStream_t s = (hipStream_t)stream;
    return dtype == DT_BF16 ? dispatch_tave<__bf16>(a, D, io16, s) : dispatch_tave<_Float16>(a, D, io16, s);
}

extern "C" int otter_text_aware_pool_fwd(int dtype, const void* patches, long unit_stride, const float* text, const void* pos_packed,
                                         const float* temperature, float* out, int n_obs, int cams, int P, int D, int K, int t_rep,
                                         void* stream) {
    return tave_entry("otter_text_aware_pool_fwd", dtype, patches, unit_stride, text, pos_packed, temperature, out, n_obs, cams, P, D, K, t_rep, 0, stream);
}

extern "C" int otter_text_aware_pool_fwd_io(int dtype, const void* patches, long unit_stride, const void* text, const void* pos_packed,
                                            const float* temperature, void* out, int n_obs, int cams, int P, int D, int K, int t_rep,
                                            int io16, void* stream) {
    return tave_entry("otter_text_aware_pool_fwd_io", dtype, patches, unit_stride, text, pos_packed, temperature, out, n_obs, cams, P, D, K, t_rep, io16, stream);
}

extern "C" long otter_pos_emb_packed_elems(int P, int D) { return (long)((P + 15) / 16) * 16 * D; }

extern "C" int otter_pos_emb_pack(int dtype, const float* pos, void* packed, int P, int D, void* stream) {
    OTTER_REQUIRE(dtype == DT_BF16 || dtype == DT_F16, OTTER_ERR_INVALID_ARGUMENT, "otter_pos_emb_pack: dtype %d (bf16 or fp16)", dtype);
    OTTER_REQUIRE(pos && packed && P > 0 && D > 0 && D % 128 == 0, OTTER_ERR_INVALID_ARGUMENT,
                  "otter_pos_emb_pack: bad arguments (P > 0, D %% 128 == 0)");
    const long n = otter_pos_emb_packed_elems(P, D);
    if (dtype == DT_BF16)
        hipLaunchKernelGGL(pos_pack_kernel<__bf16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pos, (__bf16*)packed, P, D, n);
    else
        hipLaunchKernelGGL(pos_pack_kernel<_Float16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pos, (_Float16*)packed, P, D, n);
    OTTER_CHECK_LAUNCH("otter_pos_emb_pack");
    return OTTER_OK;
}
