// Fused text-aware visual extraction (memory-bound; one pass over the patch tokens).
//
// Per unit (observation m, camera c):   patches X (P x D, bf16 or fp16, un-normalised), text T (K<=16 rows x D, fp32)
//   S    = T^ . X^^T              cosine similarity (T^, X^ = L2-normalised rows)
//   A    = softmax_over_patches(S / clamp(temperature, 0, 100))
//   out  = A . (X^ + pos_emb)     (K x D, fp32)
//
// One workgroup (4 waves) per unit, one workgroup per CU.  The patch tokens stream HBM -> LDS exactly once, in
// 16-patch tiles by LDS-DMA (global_load_lds_dwordx4) through three buffers, so two tiles are in flight behind
// the one being consumed and the only wait per tile is a counted s_waitcnt.  Per tile:
//   * S^T tile (patch x text) and the Gram diagonal |x|^2 come from v_mfma_f32_16x16x32 over D; each wave
//     reduces its quarter of D and the four partial tiles are summed through LDS (identical softmax state in
//     all waves afterwards);
//   * online softmax over patches (running max / sum per text token, one text token per lane);
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

namespace {

struct TaveArgs {
    const void* patches;        // unit u at patches + u*unit_stride, [P][D] row-major, bf16 or fp16
    long unit_stride;
    const float* text;          // [n_text][K][D]
    const void* posF;           // [cams][P/16][4 waves][NDT/2][64 lanes][8]: pos_emb in MFMA-fragment order (otter_pos_emb_pack)
    const float* temperature;   // [cams]
    float* out;                 // [cams][M][K][D]
    int units, cams, P, D, K, t_rep, M;
};

// chunk swizzle: bijection on 4 bits with f(r)>>1 injective on each aligned octet of rows and
// f({0..3,12..15}) closed under ^1  (see header comment)
__device__ __forceinline__ int f_swz(int r) {
    const int lo = r & 7, hi = (r >> 3) & 1;
    return (((lo + 4 * hi) & 7) << 1) | hi;
}

template <typename T, int NDT>   // D = NDT * 64; each wave owns NDT d-tiles of 16 columns
__global__ __launch_bounds__(256) void tave_kernel(TaveArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Half16<T>::v8 v8;
    typedef typename Half16<T>::v4 v4;
    constexpr int D = NDT * 64;
    constexpr int ROWB = D * 2;                 // bytes per LDS row
    constexpr int TILE_B = 16 * ROWB;           // one 16-row image
    constexpr int CPR = D / 8;                  // 16-byte chunks per row
    constexpr int NLD = TILE_B / (256 * 16);    // LDS-DMA instructions per thread per tile
    constexpr int NST = 3;                      // patch-tile stages: two tiles in flight behind the one being consumed
    constexpr int KSW = D / 32 / 4;             // K-steps (32 columns) of the similarity product per wave
    char* sT = smem;
    char* sX = smem + TILE_B;
    float* sP = (float*)(smem + (1 + NST) * TILE_B);       // [2][4 waves][64 lanes][8]: partial S^T and Gram tiles

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int unit = blockIdx.x;
    const int m = unit / p.cams, cam = unit % p.cams;
    const int P = p.P, nt = P >> 4;
    const T* X = (const T*)p.patches + (long)unit * p.unit_stride;
    const T* posF = (const T*)p.posF + (long)cam * P * D;

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
        char* dst = sX + buf * TILE_B;
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(base + src_off[i]), LDS_PTR(dst + (wave * NLD + i) * 1024), 16, 0, 0);
    };
    stage(0, 0);
    if (nt > 1) stage(1, 1);

    // ---- text rows: L2-normalise in fp32, store as 16-bit (swizzled) ; rows >= K are zero ----
    {
        const int b = m / p.t_rep;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {                 // unrolled: the loads of the wave's four rows are in flight together
            const int r = wave * 4 + rr;
            const float* src = p.text + ((long)b * p.K + r) * D;
            float sq = 0.f;
            f32x4 v[(D / 4 + 63) / 64];
#pragma unroll
            for (int i = 0; i < (D / 4 + 63) / 64; ++i) {
                const int idx = i * 64 + lane;
                v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (idx < D / 4 && r < p.K) v[i] = *(const f32x4*)(src + idx * 4);
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
    const float temp = fminf(fmaxf(p.temperature[cam], 0.f), 100.f);
    const float inv_temp = 1.0f / temp;

    f32x4 o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float run_max = -INFINITY, run_sum = 0.f;

    const int frag_row = li * ROWB;
    const int fsw = f_swz(li);
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
        s16x8 pt[NDT / 2];
#pragma unroll
        for (int dp = 0; dp < NDT / 2; ++dp) {
            const T* src = posF + ((long)t * 4 + wave) * (NDT * 256) + (dp * 64 + lane) * 8;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pt[dp]) : "v"(src) : "memory");
        }
        if constexpr (PREFETCH) stage((t + 2) % NST, t + 2);

        // ---- partial S^T (patch x text) and Gram diagonal over this wave's quarter of D ----
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, gd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KSW; ++kk) {
            const int ks = wave * KSW + kk;
            const int coff = (((ks * 4 + g) ^ fsw) << 4);
            const v8 xf = *(const v8*)(xb + frag_row + coff);
            const v8 tf = *(const v8*)(sT + frag_row + coff);
            s = Half16<T>::mfma32(xf, tf, s);
            gd = Half16<T>::mfma32(xf, xf, gd);
        }
        float* pp = sP + (t & 1) * 2048;
        *(f32x4*)(pp + (wave * 64 + lane) * 8) = s;
        *(f32x4*)(pp + (wave * 64 + lane) * 8 + 4) = gd;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) s += *(const f32x4*)(pp + (w * 64 + lane) * 8);
        // |x|^2 of patch 4g+r sits on lane (g, li = 4g+r), Gram register r, of every wave's partial
        float a1[4], a2[4];
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float n2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) n2 += pp[(w * 64 + g * 16 + 4 * g + r) * 8 + 4 + r];
            const float rn = 1.0f / sqrtf(n2);
            a1[r] = rn;
            a2[r] = s[r] * rn * inv_temp;                 // logit of (patch 4g+r, text li)
            tmax = fmaxf(tmax, a2[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float new_max = fmaxf(run_max, tmax);
        const float alpha = __expf(run_max - new_max);     // first tile: exp(-inf) = 0
        run_max = new_max;
        float psum = 0.f;
        s16x4 b1, b2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pr = __expf(a2[r] - new_max);
            psum += pr;
            b1[r] = bits16(Half16<T>::from(pr * a1[r]));
            b2[r] = bits16(Half16<T>::from(pr));
        }
        run_sum = run_sum * alpha + psum;                   // per-lane partial (its own 4 patches per tile)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[dt] *= alpha;

        // ---- O^T += X^T (p/|x|) + pos^T p ----
        // the pos^T loads are older than the NLD prefetch instructions issued after them (in-order retirement);
        // the "+v" operands tie every later use of pt[] to this wait
#pragma unroll
        for (int dp = 0; dp < NDT / 2; ++dp) {
            if constexpr (PREFETCH) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(pt[dp]) : "n"(NLD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(pt[dp]) : : "memory");
        }
        // X^T fragments by ds_read_b64_tr_b16, also from inline asm: after an LDS-DMA the compiler puts vmcnt(0) in
        // front of the builtin form of this read (it cannot tell that the buffers differ), draining the prefetch
        s16x4 xt[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            const int d0 = (wave * NDT + dt) * 16;
            const int chunk = (d0 >> 3) + ((li & 3) >> 1);
            const unsigned addr = (unsigned)(uintptr_t)LDS_PTR(xb + tr_row * ROWB + ((chunk ^ tr_fsw) << 4) + tr_sub);
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(xt[dt]) : "v"(addr) : "memory");
        }
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xt[dt]) : : "memory");
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
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
        for (; t + 2 < nt; ++t) tile_step(t, Y{}, Y{});
        if (t + 1 < nt) { tile_step(t, N{}, Y{}); ++t; }
        if (t < nt) tile_step(t, N{}, N{});
    }

    run_sum += __shfl_xor(run_sum, 16, 64);
    run_sum += __shfl_xor(run_sum, 32, 64);
    const float inv = 1.0f / run_sum;
    if (li < p.K) {
        float* dst = p.out + (((long)cam * p.M + m) * p.K + li) * D;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            const int d0 = (wave * NDT + dt) * 16;
            *(f32x4*)(dst + d0 + 4 * g) = o[dt] * inv;
        }
    }
}

template <typename T, int NDT>
int launch_tave(const TaveArgs& a, hipStream_t s) {
    constexpr int lds = 4 * 16 * NDT * 64 * 2 + 2 * 4 * 64 * 8 * 4;       // text image + 3 patch tiles + partial-tile exchange
    auto kern = tave_kernel<T, NDT>;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            otter_set_error("otter_text_aware_pool_fwd: cannot reserve %d bytes of LDS: %s", lds, hipGetErrorString(e));
            return OTTER_ERR_LAUNCH;
        }
        configured = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.units), dim3(256), lds, s, a);
    OTTER_CHECK_LAUNCH("otter_text_aware_pool_fwd");
    return OTTER_OK;
}

// pos_emb [P][D] fp32 -> 16-bit table in the order tave_kernel's lanes consume it:
//   packed[((t*4 + w)*(NDT/2) + dp)*64 + lane][h*4 + j] = pos[16t + 4g + j][(w*NDT + 2dp + h)*16 + li],  g = lane>>4, li = lane&15
// (setup-time kernel, one thread per element)
template <typename T>
__global__ void pos_pack_kernel(const float* pos, T* packed, int P, int D) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)P * D) return;
    const int ndt = D / 64;
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    long rest = i >> 9;
    const int dp = (int)(rest % (ndt / 2)); rest /= (ndt / 2);
    const int w = (int)(rest & 3), t = (int)(rest >> 2);
    const int h = e >> 2, j = e & 3, g = lane >> 4, li = lane & 15;
    packed[i] = Half16<T>::from(pos[(long)(16 * t + 4 * g + j) * D + (w * ndt + 2 * dp + h) * 16 + li]);
}

template <typename T>
int dispatch_tave(const TaveArgs& a, int D, hipStream_t s) {
    switch (D) {
        case 128: return launch_tave<T, 2>(a, s);
        case 512: return launch_tave<T, 8>(a, s);
        case 768: return launch_tave<T, 12>(a, s);
        case 1024: return launch_tave<T, 16>(a, s);
    }
    otter_set_error("otter_text_aware_pool_fwd: D=%d unsupported (128, 512, 768, 1024)", D);
    return OTTER_ERR_UNSUPPORTED_SHAPE;
}

}  // namespace

extern "C" int otter_text_aware_pool_fwd(int dtype, const void* patches, long unit_stride, const float* text, const void* pos_packed,
                                         const float* temperature, float* out, int n_obs, int cams, int P, int D, int K, int t_rep,
                                         void* stream) {
    OTTER_REQUIRE(dtype == DT_BF16 || dtype == DT_F16, OTTER_ERR_INVALID_ARGUMENT, "otter_text_aware_pool_fwd: dtype %d (bf16 or fp16)", dtype);
    OTTER_REQUIRE(patches && text && pos_packed && temperature && out, OTTER_ERR_INVALID_ARGUMENT, "otter_text_aware_pool_fwd: null operand");
    if (n_obs == 0) return OTTER_OK;
    OTTER_REQUIRE(n_obs > 0 && cams > 0 && t_rep > 0, OTTER_ERR_INVALID_ARGUMENT, "otter_text_aware_pool_fwd: bad sizes");
    OTTER_REQUIRE(n_obs % t_rep == 0, OTTER_ERR_INVALID_ARGUMENT, "otter_text_aware_pool_fwd: n_obs=%d not a multiple of t_rep=%d", n_obs, t_rep);
    OTTER_REQUIRE(K >= 1 && K <= 16, OTTER_ERR_UNSUPPORTED_SHAPE, "otter_text_aware_pool_fwd: K=%d text tokens, supports 1..16", K);
    OTTER_REQUIRE(P >= 16 && P % 16 == 0, OTTER_ERR_UNSUPPORTED_SHAPE, "otter_text_aware_pool_fwd: P=%d must be a multiple of 16", P);
    OTTER_REQUIRE(aligned16(patches) && aligned16(text) && aligned16(pos_packed) && aligned16(out), OTTER_ERR_ALIGNMENT,
                  "otter_text_aware_pool_fwd: operands must be 16-byte aligned");
    OTTER_REQUIRE((long)n_obs * cams < (1L << 31), OTTER_ERR_UNSUPPORTED_SHAPE, "otter_text_aware_pool_fwd: grid too large");
    OTTER_REQUIRE(unit_stride >= (long)P * D && unit_stride % 8 == 0, OTTER_ERR_ALIGNMENT,
                  "otter_text_aware_pool_fwd: unit_stride=%ld must be >= P*D and a multiple of 8 elements", unit_stride);
    TaveArgs a{patches, unit_stride, text, pos_packed, temperature, out, n_obs * cams, cams, P, D, K, t_rep, n_obs};
    hipStream_t s = (hipStream_t)stream;
    return dtype == DT_BF16 ? dispatch_tave<__bf16>(a, D, s) : dispatch_tave<_Float16>(a, D, s);
}

extern "C" int otter_pos_emb_pack(int dtype, const float* pos, void* packed, int P, int D, void* stream) {
    OTTER_REQUIRE(dtype == DT_BF16 || dtype == DT_F16, OTTER_ERR_INVALID_ARGUMENT, "otter_pos_emb_pack: dtype %d (bf16 or fp16)", dtype);
    OTTER_REQUIRE(pos && packed && P > 0 && D > 0 && P % 16 == 0 && D % 128 == 0, OTTER_ERR_INVALID_ARGUMENT,
                  "otter_pos_emb_pack: bad arguments (P %% 16 == 0, D %% 128 == 0)");
    const long n = (long)P * D;
    if (dtype == DT_BF16)
        hipLaunchKernelGGL(pos_pack_kernel<__bf16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pos, (__bf16*)packed, P, D);
    else
        hipLaunchKernelGGL(pos_pack_kernel<_Float16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pos, (_Float16*)packed, P, D);
    OTTER_CHECK_LAUNCH("otter_pos_emb_pack");
    return OTTER_OK;
}
