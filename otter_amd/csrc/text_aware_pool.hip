// Fused text-aware visual extraction (memory-bound; one pass over the patch tokens).
//
// Per unit (observation m, camera c):   patches X (P x D, bf16, un-normalised), text T (K<=16 rows x D, fp32)
//   S    = T^ . X^^T              cosine similarity (T^, X^ = L2-normalised rows)
//   A    = softmax_over_patches(S / clamp(temperature, 0, 100))
//   out  = A . (X^ + pos_emb)     (K x D, fp32)
//
// One workgroup (4 waves) per unit, 2 workgroups per CU.  The patch tokens stream HBM -> LDS exactly once, in
// 16-patch tiles by LDS-DMA (global_load_lds_dwordx4) into two buffers, so the next tile is in flight while
// the current one is consumed.  Per tile:
//   * S^T tile (patch x text) and the Gram diagonal |x|^2 come from v_mfma_f32_16x16x32_bf16 over D
//     (every wave computes the same 16x16 tile: no cross-wave exchange, identical softmax state in all waves);
//   * online softmax over patches (running max / sum per text token, one text token per lane);
//   * O^T += X^T . (p/|x|)  +  pos^T . p   with v_mfma_f32_16x16x16_bf16: the S^T accumulator layout
//     (patch 4g+r on the registers, text token on the lane) is exactly that instruction's B operand, X^T comes
//     from the row-major LDS tile by ds_read_b64_tr_b16, and pos^T is a pre-transposed bf16 table read from L2.
//     Each wave owns D/4 output columns.
// LDS rows are D*2 bytes (a multiple of 256), so the 16-byte chunk index is XORed with f(row) (below) which
// makes both the ds_read_b128 fragment reads and the transposed reads bank-conflict free.  The LDS image is
// lane-linear (LDS-DMA requirement); the swizzle is applied to the per-lane source address.
#include "common.hpp"
#include "otter_hip.h"

namespace {

struct TaveArgs {
    const void* patches;        // unit u at patches + u*unit_stride, [P][D] row-major, bf16 or fp16
    long unit_stride;
    const float* text;          // [n_text][K][D]
    const void* posT;           // [cams][D][P]  (pos_emb transposed, same 16-bit type as the patches)
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
__global__ __launch_bounds__(256, 2) void tave_kernel(TaveArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Half16<T>::v8 v8;
    typedef typename Half16<T>::v4 v4;
    constexpr int D = NDT * 64;
    constexpr int ROWB = D * 2;                 // bytes per LDS row
    constexpr int TILE_B = 16 * ROWB;           // one 16-row image
    constexpr int CPR = D / 8;                  // 16-byte chunks per row
    constexpr int NLD = TILE_B / (256 * 16);    // LDS-DMA instructions per wave per tile
    char* sT = smem;
    char* sX = smem + TILE_B;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int unit = blockIdx.x;
    const int m = unit / p.cams, cam = unit % p.cams;
    const int P = p.P, nt = P >> 4;
    const T* X = (const T*)p.patches + (long)unit * p.unit_stride;
    const T* posT = (const T*)p.posT + (long)cam * D * P;

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

    // ---- text rows: L2-normalise in fp32, store bf16 (swizzled) ; rows >= K are zero ----
    {
        const int b = m / p.t_rep;
        for (int rr = 0; rr < 4; ++rr) {
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

    __syncthreads();   // tile 0 and the text image are in LDS

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) stage(cur ^ 1, t + 1);
        const char* xb = sX + cur * TILE_B;

        // ---- S^T (patch x text) and Gram diagonal over D ----
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, gd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int ks = 0; ks < D / 32; ++ks) {
            const int coff = (((ks * 4 + g) ^ fsw) << 4);
            const v8 xf = *(const v8*)(xb + frag_row + coff);
            const v8 tf = *(const v8*)(sT + frag_row + coff);
            s = Half16<T>::mfma32(xf, tf, s);
            gd = Half16<T>::mfma32(xf, xf, gd);
        }
        // |x|^2 of patch 4g+r sits on lane (g, li = 4g+r), register r
        float a1[4], a2[4];
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float n2 = __shfl(gd[r], g * 16 + 4 * g + r, 64);
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
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            const int d0 = (wave * NDT + dt) * 16;
            const int chunk = (d0 >> 3) + ((li & 3) >> 1);
            const s16x4 xt = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(xb + tr_row * ROWB + ((chunk ^ tr_fsw) << 4) + tr_sub));
            const s16x4 pt = *(const s16x4*)(posT + (long)(d0 + li) * P + t * 16 + 4 * g);
            o[dt] = Half16<T>::mfma16(xt, b1, o[dt]);
            o[dt] = Half16<T>::mfma16(pt, b2, o[dt]);
        }
        __syncthreads();   // next tile landed (vmcnt(0)) and every wave is done with this buffer
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
    constexpr int lds = 3 * 16 * NDT * 64 * 2;
    auto kern = tave_kernel<T, NDT>;
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        configured = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.units), dim3(256), lds, s, a);
    OTTER_CHECK_LAUNCH("otter_text_aware_pool_fwd");
    return OTTER_OK;
}

template <typename T>
__global__ void pos_transpose_kernel(const float* pos, T* posT, int P, int D) {
    // posT[d][k] = T(pos[k][d]); one thread per output element (setup-time kernel)
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)P * D) return;
    const int d = (int)(i / P), k = (int)(i % P);
    posT[i] = Half16<T>::from(pos[(long)k * D + d]);
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

extern "C" int otter_text_aware_pool_fwd(int dtype, const void* patches, long unit_stride, const float* text, const void* posT, const float* temperature,
                                         float* out, int n_obs, int cams, int P, int D, int K, int t_rep, void* stream) {
    OTTER_REQUIRE(dtype == DT_BF16 || dtype == DT_F16, OTTER_ERR_INVALID_ARGUMENT, "otter_text_aware_pool_fwd: dtype %d (bf16 or fp16)", dtype);
    OTTER_REQUIRE(patches && text && posT && temperature && out, OTTER_ERR_INVALID_ARGUMENT, "otter_text_aware_pool_fwd: null operand");
    if (n_obs == 0) return OTTER_OK;
    OTTER_REQUIRE(n_obs > 0 && cams > 0 && t_rep > 0, OTTER_ERR_INVALID_ARGUMENT, "otter_text_aware_pool_fwd: bad sizes");
    OTTER_REQUIRE(n_obs % t_rep == 0, OTTER_ERR_INVALID_ARGUMENT, "otter_text_aware_pool_fwd: n_obs=%d not a multiple of t_rep=%d", n_obs, t_rep);
    OTTER_REQUIRE(K >= 1 && K <= 16, OTTER_ERR_UNSUPPORTED_SHAPE, "otter_text_aware_pool_fwd: K=%d text tokens, supports 1..16", K);
    OTTER_REQUIRE(P >= 16 && P % 16 == 0, OTTER_ERR_UNSUPPORTED_SHAPE, "otter_text_aware_pool_fwd: P=%d must be a multiple of 16", P);
    OTTER_REQUIRE(aligned16(patches) && aligned16(text) && aligned16(posT) && aligned16(out), OTTER_ERR_ALIGNMENT,
                  "otter_text_aware_pool_fwd: operands must be 16-byte aligned");
    OTTER_REQUIRE((long)n_obs * cams < (1L << 31), OTTER_ERR_UNSUPPORTED_SHAPE, "otter_text_aware_pool_fwd: grid too large");
    OTTER_REQUIRE(unit_stride >= (long)P * D && unit_stride % 8 == 0, OTTER_ERR_ALIGNMENT,
                  "otter_text_aware_pool_fwd: unit_stride=%ld must be >= P*D and a multiple of 8 elements", unit_stride);
    TaveArgs a{patches, unit_stride, text, posT, temperature, out, n_obs * cams, cams, P, D, K, t_rep, n_obs};
    hipStream_t s = (hipStream_t)stream;
    return dtype == DT_BF16 ? dispatch_tave<__bf16>(a, D, s) : dispatch_tave<_Float16>(a, D, s);
}

extern "C" int otter_pos_emb_transpose(int dtype, const float* pos, void* posT, int P, int D, void* stream) {
    OTTER_REQUIRE(dtype == DT_BF16 || dtype == DT_F16, OTTER_ERR_INVALID_ARGUMENT, "otter_pos_emb_transpose: dtype %d (bf16 or fp16)", dtype);
    OTTER_REQUIRE(pos && posT && P > 0 && D > 0, OTTER_ERR_INVALID_ARGUMENT, "otter_pos_emb_transpose: bad arguments");
    const long n = (long)P * D;
    if (dtype == DT_BF16)
        hipLaunchKernelGGL(pos_transpose_kernel<__bf16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pos, (__bf16*)posT, P, D);
    else
        hipLaunchKernelGGL(pos_transpose_kernel<_Float16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pos, (_Float16*)posT, P, D);
    OTTER_CHECK_LAUNCH("otter_pos_emb_transpose");
    return OTTER_OK;
}
