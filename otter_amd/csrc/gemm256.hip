// 256x256-tile MFMA GEMM for the large CLIP-tower shapes (bf16 / fp16 operands, fp32 accumulation).
//
//   C[M,N] = epi(A[M,K] * W[N,K]^T + bias [+ residual])     same contract as the 128x128 kernel in gemm.hip
//
// One workgroup = 8 waves (2 along M x 4 along N) works on one 256x256 output tile at a time, one persistent
// workgroup per CU looping over tiles (M must be a multiple of 256: the caller routes remainder rows to the
// 64x64 kernel); each wave
// owns 128(M) x 64(N) = 8x4 accumulator tiles of 16x16 (128 accumulator registers).  A K-step is 64 elements
// (128 bytes per row).
//
// Data movement.  Each operand tile of a K-step is cut into two 16 KiB "half-tiles" (128 rows x 128 bytes):
//   A_h0 = tile rows {0..63, 128..191}   A_h1 = tile rows {64..127, 192..255}
//   B_h0, B_h1 = the two 32-row halves of every wave column's 64 weight rows (which output columns those are: a_th / w_th)
// so that every wave's work splits into four quadrants (A_h0,B_h0) (A_h0,B_h1) (A_h1,B_h1) (A_h1,B_h0), 16 MFMAs each.
// A K-step runs as two halves, one per A half-tile: a load segment (fragment reads of the A half -- and, in the
// first half, of both B halves, which stay in registers -- plus LDS-DMA issues) and an MFMA segment of two quadrants,
// separated by workgroup barriers (four per K-step).
// LDS holds two K-steps x four half-tile slots (128 KiB).  Half-tiles arrive by LDS-DMA (global_load_lds_dwordx4,
// two instructions per thread per half-tile) and a slot is refilled as soon as its last reader segment is over:
//   first load segment of K-step s: A_h1(s+1)      second: A_h0(s+2), B_h0(s+2), B_h1(s+2)
// Every load therefore has more than a K-step (>= 2048 MFMA cycles) to land, and only ONE counted wait per K-step is
// needed: s_waitcnt vmcnt(6) at the end of the K-step leaves the six youngest requests in flight and guarantees
// that all of K-step s+1 has landed (the barrier that follows publishes it to the other waves).  The loop never
// drains the memory queue; raw s_barrier is used because __syncthreads() would add vmcnt(0).
//
// The K-steps of all the tiles a workgroup computes form ONE stream: s counts across tile boundaries, so the loads of
// the next tile's first two K-steps are requested from the last two K-steps of the current tile (whose slots are free
// by the same rule as inside a tile), land under its MFMAs and its epilogue, and the next tile's K-loop starts without a
// prologue, a drain of the memory queue or a barrier of its own.  The epilogue's stores sit in the queue between the
// loads of the two K-steps and are first waited for by the counted wait at the end of the next tile's first K-step.
// Every K-step makes the same eight requests, also the last two of the stream (which re-request the stream's last K-step
// into slots nobody reads again): with the same number of requests on every path the counted waits -- the hand-written
// ones and those the compiler places in front of the epilogue's operands -- never have to fall back to draining the queue.
//
// LDS images are lane-linear (LDS-DMA requirement) with the 16-byte chunk index XOR-swizzled on the source
// address and on the ds_read_b128 address, as in gemm.hip.
#include <type_traits>

#include "common.hpp"
#include "otter_hip.h"

namespace {

constexpr int TM = 256, TN = 256;
constexpr int HALF_BYTES = 128 * 128;             // one half-tile
constexpr int STAGE_BYTES = 2048;                 // per wave: epilogue operands staged through LDS (see stage_fetch)
constexpr int LDS_BYTES = 2 * 4 * HALF_BYTES + 8 * STAGE_BYTES;     // 128 KiB of tiles + 16 KiB
// Tile order: groups of `group_m` row tiles swept along N (Gemm256Args::group_m, chosen per shape on the host).  Measured in the
// step, where the operands of a launch come from memory (a loop over one launch keeps them in the caches and shows no
// difference): 8 for outputs of up to 12 column tiles (against 2, 4, 6, 16, 32 on the ViT-L shapes), 6 for 16 column tiles
// (fc1, N = 4096: step -0.6 % against 8; 4 and 5 -0.45 %, 7 -0.3 %).
// Outputs of at most four column tiles (out_proj, fc2) take the tile order as it is instead of one contiguous chunk per XCD:
// workgroups go to the XCDs round-robin, so with groups of 8 row tiles XCD x gets row tile x of every group with all its
// column tiles -- one A panel and the whole (narrow) weight in its L2 (step -0.45 % in four same-box rounds, all of it from
// out_proj; for the wide outputs the chunked map wins by 0.4-0.9 % of the step per shape).
constexpr int GROUP_M256 = 8, GROUP_M256_WIDE = 6;

struct Gemm256Args {
    const char* A; long lda;
    const char* W;
    const float* bias;
    const void* residual; long ldr;     // in the OUTPUT's type: fp32 with OUT_F32, else the 16-bit operand type
    void* C; long ldc;
    int M, N, K, tiles_m, tiles_n;
    int group_m;                // row tiles per group of the tile order
    int xcd_chunks;             // 1: each XCD works on one contiguous chunk of the tile order; 0: workgroup b starts at tile b
    const float* ln_stats;      // [M][2] (mean, rstd) or null: LayerNorm folded into the epilogue (see otter_gemm_ln)
    const float* ln_colsum;     // [N]
};

enum { SLOT_A0 = 0, SLOT_B0 = 1, SLOT_B1 = 2, SLOT_A1 = 3 };
enum { ACT_NONE = 0, ACT_QUICKGELU = 1, ACT_GELU = 2, ACT_RELU = 3 };

template <int ACT> __device__ __forceinline__ float act256(float v) {
    if (ACT == ACT_QUICKGELU) return quick_gelu(v);
    if (ACT == ACT_GELU) return gelu_erf_fast(v);
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    return v;
}

__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }

// LNF: LayerNorm folded into the epilogue (p.ln_stats).
template <typename T, int ACT, bool OUT_F32, bool RES, bool LNF = false>
__global__ __launch_bounds__(512, 1) void gemm_256x256(Gemm256Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Half16<T>::v8 frag_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: everything derived from it is scalar
    const int wr = wave >> 2, wc = wave & 3;
    const int g = lane >> 4, li = lane & 15;
    fp16_saturate_on_overflow();

    const int nwg = p.tiles_m * p.tiles_n;
    const int nkt = p.K / 64;
    // LDS-DMA source offsets of this thread inside a tile's row blocks (bytes from the tile's first A row / first weight
    // row, K-step 0): [half][instruction].  They do not depend on the tile: a request's address is a scalar base (tile
    // origin + K-step) plus one of these.
    // The accumulator of an (m-tile, n-tile) pair holds C[16mt + 4g + r][column li of the n-tile] in register r of lane
    // (g, li).  Image row (n-tile nt, i) of half x of a wave's 64 columns is given the weight row of column 4i + 2x + nt, so
    // the four n-tiles of a lane are 4 CONSECUTIVE output columns: the 16 lanes of a group store one contiguous 128-byte
    // (16-bit output) or 256-byte (fp32) piece of a row, every quad of lanes a contiguous part of it (stores whose quads
    // touch four rows take four times as long in the address unit).
    unsigned a_th[2][2], w_th[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int lr = (wave * 2 + i) * 8 + (lane >> 3);            // local row inside the half-tile image
        const int c = (lane & 7) ^ ((lr >> 1) & 7);
        const int fi = lr & 15, fnt = (lr >> 4) & 1;
        const int wcol = (lr >> 5) * 64 + 4 * fi + fnt;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int trow = (lr < 64 ? lr : lr + 64) + x * 64;      // A: tile row
            a_th[x][i] = (unsigned)trow * (unsigned)(p.lda * 2) + c * 16;
            w_th[x][i] = (unsigned)(wcol + x * 2) * (unsigned)(p.K * 2) + c * 16;
        }
    }
    // Tile `t` of the launch order -> (m0, n0); scalar arithmetic
    auto tile_origin = [&](int t, int& m0, int& n0) {
        const int pid = p.xcd_chunks ? xcd_remap(t, nwg) : t;
        const int width = p.group_m * p.tiles_n;
        const int first_m = (pid / width) * p.group_m;
        const int gsz = min(p.tiles_m - first_m, p.group_m);
        m0 = (first_m + (pid % width) % gsz) * TM;
        n0 = ((pid % width) / gsz) * TN;
    };
    // (32-bit offsets from A and W: the request's address is a scalar 64-bit base plus a 32-bit per-lane offset)
    auto issue = [&](int buf, int slot, unsigned a_base, unsigned w_base) {
        char* dst = smem + (buf * 4 + slot) * HALF_BYTES + wave * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const char* src;
            if (slot == SLOT_A0) src = p.A + (a_base + a_th[0][i]);
            else if (slot == SLOT_A1) src = p.A + (a_base + a_th[1][i]);
            else if (slot == SLOT_B0) src = p.W + (w_base + w_th[0][i]);
            else src = p.W + (w_base + w_th[1][i]);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(dst + i * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read offsets inside a half-tile image ----
    const int swz = (li >> 1) & 7;
    int koff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) koff[ks] = li * 128 + (((ks * 4 + g) ^ swz) << 4);
    const int a_row0 = wr * 64 * 128;          // this wave's 64 rows inside an A half-tile
    const int b_row0 = wc * 32 * 128;          // this wave's 32 rows inside a B half-tile

    f32x4 acc[2][2][4][2];                      // [A half s][B half x][m-tile][n-tile]
    const bool late = wave >= 4;

    // ---- the load stream (wave-uniform state, scalar registers).  Cursor 2 = the K-step two ahead of the one being
    // computed (its A_h0, B_h0, B_h1 are requested next), cursor 1 = the K-step one ahead (its A_h1 is requested next).
    // Advancing a cursor is a handful of scalar selects without a branch: it runs between a K-step's last barrier and the
    // next load segment, which the other wave group's MFMA segment waits for.  The one expensive part, a tile's origin (a
    // division), is computed once per tile after the epilogue's stores, a tile ahead of the stream: `nx_*` is the tile the
    // stream enters at its next wrap.  (The stream wraps in K-step nkt - 3 of the tile being computed: nkt >= 4.)
    int tile = blockIdx.x;                      // the tile being computed
    int cm0, cn0;                               // its origin
    tile_origin(tile, cm0, cn0);
    unsigned a2 = (unsigned)cm0 * (unsigned)(p.lda * 2);         // byte offsets of the K-step's first A row / weight row
    unsigned w2 = (unsigned)cn0 * (unsigned)(p.K * 2);
    int s_k = 0;                                // K-step of cursor 2 inside its tile
    unsigned s_inc = 1, s_step = 128;           // both 0 once the stream has ended: the cursor stays on the last K-step it
                                                // reached (requests keep being made -- see k_step -- and must name memory that exists)
    int nx_tile, nx_m0 = 0, nx_n0 = 0;          // the stream's next tile; its origin (valid while nx_tile < nwg)
    int cnx_m0 = 0, cnx_n0 = 0;                 // origin of the tile computed next (= the tile the stream entered last)
    auto prepare_next = [&](int t) {
        nx_tile = t;
        if (t < nwg) tile_origin(t, nx_m0, nx_n0);
    };
    prepare_next(tile + gridDim.x);
    auto advance2 = [&]() {
        s_k += s_inc;
        const bool wrap = s_k == nkt;
        const bool more = nx_tile < nwg;
        const bool enter = wrap && more, end = wrap && !more;
        a2 = enter ? (unsigned)nx_m0 * (unsigned)(p.lda * 2) : a2 + (wrap ? 0u : s_step);
        w2 = enter ? (unsigned)nx_n0 * (unsigned)(p.K * 2) : w2 + (wrap ? 0u : s_step);
        cnx_m0 = enter ? nx_m0 : cnx_m0;
        cnx_n0 = enter ? nx_n0 : cnx_n0;
        s_k = enter ? 0 : (end ? nkt - 1 : s_k);
        s_inc = end ? 0u : s_inc;
        s_step = end ? 0u : s_step;
    };
    // K-step 0 completely, plus the three half-tiles of K-step 1 that the second half of K-step -1 would request
    issue(0, SLOT_A0, a2, w2);
    issue(0, SLOT_B0, a2, w2);
    issue(0, SLOT_B1, a2, w2);
    issue(0, SLOT_A1, a2, w2);
    advance2();
    unsigned a1 = a2;
    issue(1, SLOT_A0, a2, w2);
    issue(1, SLOT_B0, a2, w2);
    issue(1, SLOT_B1, a2, w2);
    advance2();
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    wg_barrier();

    frag_t af[4][2], bf[2][2][2];               // A half in use; both B halves of the K-step: [half x][n-tile][k-sub-step]
    auto load_a = [&](int buf, int slot, int mt0, int mt1) {
        const char* base = smem + (buf * 4 + slot) * HALF_BYTES + a_row0;
#pragma unroll
        for (int mt = mt0; mt < mt1; ++mt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[mt][ks] = *(const frag_t*)(base + mt * 2048 + koff[ks]);
    };
    auto load_b = [&](int buf, int slot, int x) {
        const char* base = smem + (buf * 4 + slot) * HALF_BYTES + b_row0;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bf[x][nt][ks] = *(const frag_t*)(base + nt * 2048 + koff[ks]);
    };
    // a tile's first K-step starts its accumulators from the constant 0 (no zero-fill pass over 128 registers)
#define OTTER_QUADRANT(S, X)                                                                              \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                       \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                   \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                               \
                acc[S][X][mt][nt] = Half16<T>::mfma32(af[mt][ks], bf[X][nt][ks],                           \
                                                      (FIRST && ks == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[S][X][mt][nt]);

    // A K-step is two halves, one per A half-tile: load segment (fragment reads + LDS-DMA requests) then MFMA segment (two
    // quadrants = 32 MFMAs per wave), four workgroup barriers per K-step.  Both B halves are read in the first half
    // and stay in registers for the second.
    // Two wave groups (waves 0-3 / 4-7; each SIMD hosts one wave of each) run the same program one barrier apart:
    // while one group is in a load segment the other is in its MFMA segment, so the matrix pipe of every SIMD always has
    // a wave feeding it.  Group 1 starts one barrier late (once, ahead of the first tile) and group 0 ends with one extra
    // barrier (after the last tile).
    // Slot refills (write after read): A_h0, B_h0, B_h1 of a buffer are last read in the first load segment, so their
    // K-step s+2 contents are requested in the second one; A_h1 is last read in the second and refilled in the first
    // load segment of K-step s+1.  K-step s+1 is therefore complete once at most the six requests of the second load
    // segment of K-step s are outstanding: one counted wait per K-step, at the end of the same global segment for
    // both groups, one barrier before the first reader.
#define OTTER_SEG_END()                                     \
    __builtin_amdgcn_sched_barrier(0);                      \
    wg_barrier();                                           \
    __builtin_amdgcn_sched_barrier(0);
#define OTTER_MFMA_SEG2(S, X0, X1)                          \
    OTTER_QUADRANT(S, X0)                                   \
    OTTER_QUADRANT(S, X1)
#define OTTER_RETIRE_NEXT() asm volatile("s_waitcnt vmcnt(6)" ::: "memory");

    // ---- epilogue operands that come from memory.  Loaded into registers where they are used, each costs the epilogue a
    // memory round trip behind the load stream's requests in flight (and the compiler, which cannot count requests across
    // the tile loop, drains the whole queue in front of them).  The small ones -- bias and LayerNorm column sums of the
    // wave's 64 columns, (mean, rstd) of its 128 rows -- are therefore requested from the tile's LAST K-step by LDS-DMA into
    // a 2 KiB staging area of the wave's own (no registers held, complete at that K-step's counted wait, read back by
    // ds_read in the epilogue):   [0, 1024) (mean, rstd) x 128 rows   [1024, 1280) bias x 64   [1280, 1536) column sums x 64
    constexpr int EO = OUT_F32 ? 4 : 2;                    // bytes per output element
    typedef typename Half16<T>::v4 h4;
    typedef typename Half16<T>::v2 h2;
    const unsigned lane_c = (unsigned)(4 * g) * (unsigned)(p.ldc * EO) + (unsigned)(4 * li * EO);
    const unsigned lane_r = (unsigned)(4 * g) * (unsigned)(p.ldr * EO) + (unsigned)(4 * li * EO);
    char* const stage = smem + 2 * 4 * HALF_BYTES + wave * STAGE_BYTES;
    auto stage_fetch = [&]() {
        const int n = cn0 + wc * 64 + lane;
        const int n_ld = n < p.N ? n : cn0;                 // lanes beyond N: any column in bounds, never used
        if (p.bias) __builtin_amdgcn_global_load_lds(GLOBAL_PTR(p.bias + n_ld), LDS_PTR(stage + 1024), 4, 0, 0);
        if (LNF) {
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(p.ln_colsum + n_ld), LDS_PTR(stage + 1280), 4, 0, 0);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(p.ln_stats + 2 * (long)(cm0 + wr * 128) + 4 * lane), LDS_PTR(stage), 16, 0, 0);
        }
    };

    int buf = 0;                                // LDS buffer of the K-step being computed (stream parity)
    auto k_step = [&](auto first_tag, bool last) {
        constexpr bool FIRST = decltype(first_tag)::value;
        // first half: (A_h0, B_h0), (A_h0, B_h1)
        load_a(buf, SLOT_A0, 0, 4);
        load_b(buf, SLOT_B0, 0);
        load_b(buf, SLOT_B1, 1);
        issue(buf ^ 1, SLOT_A1, a1, w2);
        if (last) stage_fetch();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        OTTER_SEG_END()
        OTTER_MFMA_SEG2(0, 0, 1)
        OTTER_SEG_END()
        // second half: (A_h1, B_h1), (A_h1, B_h0)
        load_a(buf, SLOT_A1, 0, 4);
        issue(buf, SLOT_A0, a2, w2);
        issue(buf, SLOT_B0, a2, w2);
        issue(buf, SLOT_B1, a2, w2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (late) { OTTER_RETIRE_NEXT() }
        OTTER_SEG_END()
        OTTER_MFMA_SEG2(1, 1, 0)
        if (!late) { OTTER_RETIRE_NEXT() }
        if (!last) { OTTER_SEG_END() }          // (a tile's last K-step: the barrier follows the epilogue's requests, below)
        a1 = a2;
        advance2();
        buf ^= 1;
    };

    if (late) {
        // static priority for the group that was dispatched second (it loses the age-based arbitration of the SIMD's issue
        // slots in every segment otherwise); no per-segment flips
        __builtin_amdgcn_s_setprio(1);
        OTTER_SEG_END()
    }
    while (true) {
    k_step(std::true_type{}, nkt == 1);
    for (int t = 1; t < nkt; ++t) k_step(std::false_type{}, t == nkt - 1);
    // 16-bit residual: all 32 rows of the lane are requested here, behind the last MFMA segment and ahead of its barrier
    // (the fragment registers are free now): they travel while this group waits at the barrier -- the late group through the
    // early group's first stores -- and no load has to queue behind a store.
    h4 rh[2][RES && !OUT_F32 ? 16 : 1];
    if constexpr (RES && !OUT_F32) {
        // lanes whose columns lie beyond N (last column tile of an N that is not a multiple of 256) read the tile's first
        // columns instead: in bounds, never used
        unsigned lo = lane_r - (unsigned)(4 * li * EO) + (cn0 + wc * 64 + 4 * li < p.N ? (unsigned)((wc * 64 + 4 * li) * EO) : 0u);
        asm volatile("" : "+v"(lo));
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const char* r_u = (const char*)p.residual + ((long)(cm0 + wr * 128 + s * 64) * p.ldr + cn0) * EO;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) rh[s][mt * 4 + r] = *(const h4*)(r_u + (long)(mt * 16 + r) * (p.ldr * EO) + (unsigned long)lo);
        }
    }
    OTTER_SEG_END()

    // ---- store the tile (the next tile's first K-steps are landing meanwhile) ----
    // lane (g, li): rows 16mt + 4g + r, columns 4li + 0..3 of the wave's 64.
    // Every address is a wave-uniform 64-bit part (tile origin, wave block, row: scalar registers and scalar arithmetic)
    // plus a 32-bit per-lane byte offset that never changes (4g rows, 4li columns), so a store or load costs no vector
    // instruction for its address; the element arithmetic is written per scalar so that the accumulators are consumed
    // where they sit (no register shuffles to build vectors) and the 16-bit conversion saturates in hardware (mode bit
    // set at kernel entry).  The vector ALU, shared with the partner wave, sets the pace of this part of a tile.
    {
        const int n_u = cn0 + wc * 64;
        const int n = n_u + 4 * li;
        const bool n_ok = n < p.N;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f}, cs = bv;
        if (p.bias) bv = *(const f32x4*)(stage + 1024 + 16 * li);
        if (LNF) cs = *(const f32x4*)(stage + 1280 + 16 * li);
        if constexpr (RES && !OUT_F32) {
            // one wait for all 32 residual rows here, where nothing younger is in flight: the compiler's wait in front of the
            // second half would otherwise also cover the first half's stores
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(rh[s][i]));
        }
        {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int row_u = cm0 + wr * 128 + s * 64;                              // wave-uniform
                char* c_u = (char*)p.C + ((long)row_u * p.ldc + n_u) * EO;
                unsigned lo_c = lane_c, lo_r = lane_r;
                asm volatile("" : "+v"(lo_c), "+v"(lo_r));
                // fp32 residual: the 8 rows of two m-tiles first (32 registers; all 16 of the half do not fit), then their stores
                f32x4 rf[RES && OUT_F32 ? 8 : 1];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    if (RES && OUT_F32 && n_ok && (mt & 1) == 0) {
                        const char* r_u = (const char*)p.residual + ((long)row_u * p.ldr + n_u) * EO;
#pragma unroll
                        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const char* src = r_u + (long)((mt + m2) * 16 + r) * (p.ldr * EO) + (unsigned long)lo_r;
                                rf[m2 * 4 + r] = *(const f32x4*)src;
                            }
                    }
                    f32x4 st2[LNF ? 2 : 1];                 // (mean, rstd) of rows r = 0, 1 and of rows r = 2, 3
                    if (LNF) {
                        st2[0] = *(const f32x4*)(stage + (s * 64 + mt * 16 + 4 * g) * 8);
                        st2[1] = *(const f32x4*)(stage + (s * 64 + mt * 16 + 4 * g) * 8 + 16);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v[4] = {acc[s][0][mt][0][r], acc[s][0][mt][1][r], acc[s][1][mt][0][r], acc[s][1][mt][1][r]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (LNF) v[e] = __builtin_fmaf(__builtin_fmaf(-st2[r >> 1][2 * (r & 1)], cs[e], v[e]), st2[r >> 1][2 * (r & 1) + 1], bv[e]);
                            else v[e] += bv[e];
                            v[e] = act256<ACT>(v[e]);
                            if (RES) {
                                if constexpr (OUT_F32) v[e] += rf[(mt & 1) * 4 + r][e];
                                else v[e] += (float)rh[s][mt * 4 + r][e];
                            }
                        }
                        char* dst = c_u + (long)(mt * 16 + r) * (p.ldc * EO) + (unsigned long)lo_c;
                        if constexpr (OUT_F32) {
                            if (n_ok) *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
                        } else {
                            union { h2 h[2]; h4 q; } o;
                            o.h[0] = Half16<T>::pack_sat_hw(v[0], v[1]);
                            o.h[1] = Half16<T>::pack_sat_hw(v[2], v[3]);
                            if (n_ok) *(h4*)dst = o.q;
                        }
                    }
                }
                // The early group stores its first 64 rows beside the late group's last MFMA segment of the tile and its
                // second 64 beside the late group's whole epilogue: two waves per SIMD share the vector ALU at twice the
                // rate one wave can issue at, so the two groups' epilogues overlap instead of following each other.  One
                // extra barrier per tile and group keeps the groups one segment apart.
                if (s == 0 && !late) { OTTER_SEG_END() }
            }
            if (late) { OTTER_SEG_END() }
        }
    }
    tile += gridDim.x;
    if (tile >= nwg) break;
    cm0 = cnx_m0;                               // the stream entered this tile during the K-loop above
    cn0 = cnx_n0;
    prepare_next(tile + gridDim.x);
    }   // tile loop
    if (!late) { OTTER_SEG_END() }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA request may outlive the workgroup's LDS allocation
#undef OTTER_SEG_END
#undef OTTER_MFMA_SEG2
#undef OTTER_RETIRE_NEXT
#undef OTTER_QUADRANT
}

template <typename T, int ACT, bool OUT_F32, bool RES, bool LNF = false>
int launch256(const Gemm256Args& a, hipStream_t s) {
    auto kern = gemm_256x256<T, ACT, OUT_F32, RES, LNF>;
    if (int rc = otter_reserve_lds((const void*)kern, LDS_BYTES, "otter_gemm (256x256)")) return rc;
    const int n_cu = otter_cu_count("otter_gemm (256x256)");
    if (n_cu <= 0) return OTTER_ERR_LAUNCH;
    const int tiles = a.tiles_m * a.tiles_n;
    hipLaunchKernelGGL(kern, dim3(tiles < n_cu ? tiles : n_cu), dim3(512), LDS_BYTES, s, a);
    OTTER_CHECK_LAUNCH("otter_gemm (256x256)");
    return OTTER_OK;
}

template <typename T>
int dispatch256(const Gemm256Args& a, int act, bool out_f32, bool res, hipStream_t s) {
    if (res) return out_f32 ? launch256<T, ACT_NONE, true, true>(a, s) : launch256<T, ACT_NONE, false, true>(a, s);
    if (a.ln_stats && !out_f32) {
        switch (act) {
            case OTTER_EPI_ACT_NONE: return launch256<T, ACT_NONE, false, false, true>(a, s);
            case OTTER_EPI_ACT_QUICKGELU: return launch256<T, ACT_QUICKGELU, false, false, true>(a, s);
            case OTTER_EPI_ACT_GELU: return launch256<T, ACT_GELU, false, false, true>(a, s);
        }
    }
    switch (act) {
        case OTTER_EPI_ACT_NONE: return out_f32 ? launch256<T, ACT_NONE, true, false>(a, s) : launch256<T, ACT_NONE, false, false>(a, s);
        case OTTER_EPI_ACT_QUICKGELU: return out_f32 ? launch256<T, ACT_QUICKGELU, true, false>(a, s) : launch256<T, ACT_QUICKGELU, false, false>(a, s);
        case OTTER_EPI_ACT_GELU: return out_f32 ? launch256<T, ACT_GELU, true, false>(a, s) : launch256<T, ACT_GELU, false, false>(a, s);
        case OTTER_EPI_ACT_RELU: return out_f32 ? launch256<T, ACT_RELU, true, false>(a, s) : launch256<T, ACT_RELU, false, false>(a, s);
    }
    otter_set_error("otter_gemm: unknown activation %d", act);
    return OTTER_ERR_INVALID_ARGUMENT;
}

}  // namespace

// Shapes the 256x256 kernel accepts (checked by the caller in gemm.hip, which owns the argument validation).
bool otter_gemm256_eligible(int dtype, long lda, int n_pad, int M, int N, int K, const void* C, long ldc) {
    if (dtype != DT_BF16 && dtype != DT_F16) return false;
    if (M < 1024 || N < 256 || K % 64 != 0 || K < 256) return false;                       // small problems: 128x128 tiles fill the chip better
    if (N % 8 != 0 || ldc % 8 != 0 || ((uintptr_t)C & 15) != 0) return false;     // a lane stores 4 consecutive columns as one vector; rows stay 16-byte aligned
    if (n_pad < ((N + TN - 1) / TN) * TN) return false;                          // weight rows must cover whole 256-column tiles
    if ((long)(M / TM) * ((N + TN - 1) / TN) < 128) return false;                // fewer tiles than half the CUs: smaller tiles fill the chip better
    if ((long)M * lda * 2 >= (1L << 32) || (long)n_pad * K * 2 >= (1L << 32)) return false;   // 32-bit source offsets
    return true;
}

int otter_gemm256_launch(int dtype, const void* A, long lda, const void* W, const float* bias, const void* residual, long ldr,
                         void* C, long ldc, int M, int N, int K, int act, bool out_f32, bool res, const float* ln_stats,
                         const float* ln_colsum, hipStream_t s) {
    if (M <= 0 || M % TM != 0) {                     // whole 256-row tiles only: the caller sends the remainder elsewhere
        otter_set_error("otter_gemm (256x256): M=%d is not a positive multiple of %d", M, TM);
        return OTTER_ERR_UNSUPPORTED_SHAPE;
    }
    const int tiles_n = (N + TN - 1) / TN;
    const int group_m = tiles_n >= 16 ? GROUP_M256_WIDE : GROUP_M256;
    const int xcd_chunks = tiles_n <= 4 ? 0 : 1;
    Gemm256Args a{(const char*)A, lda, (const char*)W, bias, residual, ldr, C, ldc, M, N, K, (M + TM - 1) / TM, tiles_n, group_m, xcd_chunks, ln_stats, ln_colsum};
    return dtype == DT_BF16 ? dispatch256<__bf16>(a, act, out_f32, res, s) : dispatch256<_Float16>(a, act, out_f32, res, s);
}
