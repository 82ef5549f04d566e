// bf16 MFMA GEMMs for the distill step (gfx950).
//
//   gemm_nt      C[M,N] = epi(alpha * A[M,K] · B[N,K]^T)      forward linears, dgrad (with pre-transposed W)
//   gemm_tn_acc  dW[P,Q] += A[M,P]^T · B[M,Q]                 wgrad (contraction over tokens), f32 atomics
//   colsum_acc   db[N]   += sum_m X[m,N]                      bias grad
//
// gemm_nt: 128x128x64 tile, 4 waves (2x2), each wave 64x64 = 4x4 tiles of v_mfma_f32_16x16x32_bf16.
// Operands go HBM -> LDS with global_load_lds (16 B/lane, no VGPR staging), double buffered, one barrier per
// K-tile.  The LDS image is made of 1 KiB [16 rows x 32 k] sub-tiles (one per wave-instruction) with the
// st_16x32 XOR swizzle (byte ^= ((byte >> 9) & 1) << 5) applied on the SOURCE address and again on the
// ds_read_b128 address (guide: cdna_hip_programming.md §5 "LDS swizzle", rule 21).

#include "gemm_common.h"
#include <atomic>

using namespace dgemm;

namespace {

constexpr int TILE_BYTES = (BM / 16) * (BK / 32) * SUB;   // 16 KiB per operand per stage
constexpr int CLD = BN + 4;                     // f32 row stride of the C tile staged through LDS in the epilogue
constexpr int NT_LDS = (4 * TILE_BYTES > BM * CLD * 4) ? 4 * TILE_BYTES : BM * CLD * 4;


// one wave stages 4 of the 16 sub-tiles of an operand tile
__device__ __forceinline__ void stage_operand(const bf16_t* __restrict__ G, int64_t ld, int row0, int rows_max,
                                              int k0, char* lds_tile, int wave, int lane) {
    const int L = lane * 16;
    const int X = swz(L);
    const int r = X >> 6, c = (X >> 4) & 3;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int sub = wave * 4 + s;          // sub = rb * 2 + kb
        const int rb = sub >> 1, kb = sub & 1;
        int row = row0 + rb * 16 + r;
        row = row < rows_max ? row : rows_max - 1;
        const bf16_t* src = G + (int64_t)row * ld + (k0 + kb * 32 + c * 8);
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(lds_tile + sub * SUB), 16, 0, 0);
    }
}

template <int ACT, int OUT>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNT p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int nwg = p.tiles_m * p.tiles_n;
    const int t = xcd_remap(blockIdx.x, nwg);
    const int tm = t / p.tiles_n, tn = t % p.tiles_n;     // n fastest: the A row-panel stays in this XCD's L2
    const int m0 = tm * BM, n0 = tn * BN;

    char* As = smem;
    char* Bs = smem + 2 * TILE_BYTES;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    stage_operand(p.A, p.lda, m0, p.M, 0, As, wave, lane);
    stage_operand(p.B, p.ldb, n0, p.N, 0, Bs, wave, lane);

    const int fragoff = swz((lane & 15) * 64 + (lane >> 4) * 16);

    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();   // tile kt landed (the fence drains vmcnt) and every wave left buffer (kt+1)&1
        if (kt + 1 < nk) {
            const int nb = (kt + 1) & 1;
            stage_operand(p.A, p.lda, m0, p.M, (kt + 1) * BK, As + nb * TILE_BYTES, wave, lane);
            stage_operand(p.B, p.ldb, n0, p.N, (kt + 1) * BK, Bs + nb * TILE_BYTES, wave, lane);
        }
        const char* a_t = As + (kt & 1) * TILE_BYTES;
        const char* b_t = Bs + (kt & 1) * TILE_BYTES;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i] = *(const bf16x8*)(a_t + ((wm * 4 + i) * 2 + kb) * SUB + fragoff);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bfr[j] = *(const bf16x8*)(b_t + ((wn * 4 + j) * 2 + kb) * SUB + fragoff);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue.  acc[i][j][r] is C[row = i*16 + (lane>>4)*4 + r][col = j*16 + (lane&15)] of the wave's 64x64 tile: storing
    // from that layout means 2-byte scattered stores.  Stage the 128x128 f32 tile through LDS (the operand buffers are
    // dead now) and run bias / activation / residual on row-contiguous 8-element vectors (16-byte loads and stores).
    __syncthreads();
    float* cs = (float*)smem;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                cs[(wm * 64 + i * 16 + (lane >> 4) * 4 + r) * CLD + wn * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
    __syncthreads();
    const int cc = (tid & 15) * 8;
    const int col = n0 + cc;
    const bool col_ok = col < p.N;
    float bias[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = 0.f;
    if (p.bias && col_ok) {
        const float4 b0 = *(const float4*)(p.bias + col), b1 = *(const float4*)(p.bias + col + 4);
        bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w; bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
    }
    float csum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) csum[e] = 0.f;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int rl = (tid >> 4) + it * 16;
        const int row = m0 + rl;
        if (row >= p.M || !col_ok) break;
        float v[8];
        {
            const float4 c0 = *(const float4*)(cs + rl * CLD + cc), c1 = *(const float4*)(cs + rl * CLD + cc + 4);
            v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w; v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * p.alpha + bias[e];
        if (p.row_group > 0) {
            const float* ra = p.rowadd + (int64_t)(row % p.row_group) * p.N + col;
            const float4 a0 = *(const float4*)ra, a1 = *(const float4*)(ra + 4);
            v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w; v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
        }
        const int64_t o = (int64_t)row * p.ldc + col;
        {   // the per-unit epilogue of the large kernels, every optional operand a run-time test (MODE 0); the side operands are loaded here
            EpiSide sd;
            epilogue_load_side<ACT, OUT>(p, o, (int64_t)row * p.ldr + col, sd);
            float zero[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, unused[8];
            GemmNT q = p;
            q.alpha = 1.f; q.row_group = 0; q.colsum = nullptr;        // (alpha, bias and the positional rows are applied above)
            epilogue_vec8<ACT, OUT, 0>(q, v, row, col, o, (int64_t)row * p.ldr + col, zero, unused, sd);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[e] += v[e];
    }
    if (p.colsum) {
        // reduce the 16 row-groups through LDS so that each atomic wave-instruction covers 256 contiguous bytes
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[(tid >> 4) * CLD + cc + e] = csum[e];
        __syncthreads();
        if (tid < BN && n0 + tid < p.N) {
            float x = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) x += cs[r * CLD + tid];
            unsafeAtomicAdd(p.colsum + n0 + tid, x);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// gemm_nt256: 256x256x64 tile (MI = 8) or 320x256x64 (MI = 10), 8 waves (2 x 4, wave tile 16 MI x 64 = MI x 4 MFMA tiles,
// 16 MI accumulator registers), one workgroup per CU (128 / 144 KiB LDS).  Structure after the guide's "256^2 8-phase"
// template, with our own schedule (round 2: two phases per K-tile, measured with in-kernel stamps, tools/diag/gemm_phases.py):
//   * LDS = [2 parities][B_lo, B_hi, A region of wave group 0, A region of wave group 1]; B halves are 128 rows x 64 k (16 KiB,
//     the swizzled [16 x 32] sub-tile image of the 128^2 kernel), an A region holds the group's MI row tiles.  The load stream
//     moves "half-tiles" staged by all 8 waves (2-3 global_load_lds per wave): B_lo, B_hi, A_up (the upper MI/2 row tiles of
//     BOTH groups), A_dn (the lower ones).
//   * a K-tile is 2 phases of 2 MI MFMAs x 2 (k halves): phase A = upper row tiles x all four column tiles (fragments: 8 B +
//     MI A), phase B = lower row tiles (MI A fragments, the B fragments stay in registers).  4 barriers per K-tile.
//   * load stream: B_lo, B_hi, A_up of tile kt + 2 are requested in phase B of tile kt, A_dn(kt + 1) in phase A of tile kt, so
//     every half-tile has two phases to land; counted s_waitcnt vmcnt(N) (vmcnt retires in issue order: N = instructions
//     issued after the half-tile that is needed), never 0 inside the loop.
//     WAR: a wave retires its own LDS reads (lgkmcnt) BEFORE the phase's first barrier, so after any barrier every read
//     issued before it has completed and the regions refilled next are idle.
//   * the two wave groups (wr = 0 / 1, one wave of each per SIMD) run one barrier apart: while one group issues its
//     MFMAs the other issues its ds_reads / LDS-DMA, so the matrix pipe of every SIMD stays fed.
//   * MFMA operands are swapped (C^T = B A^T) and the B rows permuted on the global side of the LDS-DMA, so a lane ends up
//     with 8 consecutive output columns and the epilogue stores straight from registers (stage_half_perm).
// Cycles per K-tile and workgroup (median, all CUs busy): 2 465 (MI = 8, ideal 2 048) / 2 995 (MI = 10, ideal 2 560); the
// four-phase schedule of round 1 (8 barriers, loads 5 phases ahead) ran 2 750 / 3 200, one phase per K-tile 2 770.
// ------------------------------------------------------------------------------------------------------
constexpr int HT = 16384;                       // bytes of a B half-tile

// B half-tile of the 256- / 320-row kernels: same LDS layout, but LDS row rho of sub-tile s (strip = s >> 2, j = s & 3) holds
// the operand row  strip * 64 + (j >> 1) * 32 + (rho >> 2) * 8 + (j & 1) * 4 + (rho & 3).  With the MFMA operands swapped
// (C^T = B A^T) lane (g = lane >> 4, c = lane & 15) then owns, for output row c of row tile i, the accumulators
// acc[i][2 jp][0..3], acc[i][2 jp + 1][0..3] = 8 CONSECUTIVE output columns jp * 32 + g * 8 + (0..7) of the wave's 64-column
// strip: the epilogue stores straight from registers, 16 B (bf16) / 32 B (f32) per lane, 64 / 128 contiguous bytes per row,
// with no staging through LDS and no barrier.  The permutation is applied on the global-memory side of the LDS-DMA: free.
__device__ __forceinline__ void stage_half_perm(const bf16_t* __restrict__ G, int64_t ld, int row0, int rows_max, int k0,
                                                char* buf, int wave, int lane) {
    const int L = lane * 16;
    const int X = swz(L);
    const int r = X >> 6, c = (X >> 4) & 3;
    const int j = wave & 3;
    int row = row0 + (wave >> 2) * 64 + (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
    row = row < rows_max ? row : rows_max - 1;
    const bf16_t* src = G + (int64_t)row * ld + k0 + c * 8;
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(buf + (wave * 2 + 0) * SUB), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void*)(src + 32), (lds_void*)(buf + (wave * 2 + 1) * SUB), 16, 0, 0);
}

// A "row-half" of the two-phase schedule: the upper (dn = 0) or lower (dn = 1) RH row tiles of BOTH wave groups, written to
// the same LDS addresses the fragment reads expect (group g's region at g * HTA, row tile i at sub-tiles 2 i, 2 i + 1).  Row
// blocks 0-7 by wave; with RH = 5 the remaining two row blocks are split in four [16 x 32] pieces over waves 0-3 (waves 4-7
// repeat them: every wave issues the same number of LDS-DMA instructions, so the counted vmcnt waits stay wave-independent).
template <int RH>
__device__ __forceinline__ void stage_rowhalf(const bf16_t* __restrict__ G, int64_t ld, int m0, int rows_max, int k0,
                                              char* abase, int hta, int dn, int wave, int lane) {
    const int L = lane * 16;
    const int X = swz(L);
    const int r = X >> 6, c = (X >> 4) & 3;
    constexpr int AROWS = RH * 32;
    {
        // RH = 3: six row blocks for eight waves — waves 6, 7 repeat blocks 0, 1 (same bytes to the same LDS address), so every
        // wave issues the same number of LDS-DMA instructions and the counted vmcnt waits stay wave-independent
        const int wv = RH == 3 ? wave % 6 : wave;
        const int g = wv / RH, i = wv % RH + dn * RH;              // row block `wv` of 2 RH
        int row = m0 + g * AROWS + i * 16 + r;
        row = row < rows_max ? row : rows_max - 1;
        const bf16_t* src = G + (int64_t)row * ld + k0 + c * 8;
        char* dst = abase + g * hta + (i * 2) * SUB;
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void*)(src + 32), (lds_void*)(dst + SUB), 16, 0, 0);
    }
    if (RH == 5) {
        const int e = wave & 3, rbk = 8 + (e >> 1), kb = e & 1;
        const int g = rbk / RH, i = rbk % RH + dn * RH;
        int row = m0 + g * AROWS + i * 16 + r;
        row = row < rows_max ? row : rows_max - 1;
        const bf16_t* src = G + (int64_t)row * ld + k0 + kb * 32 + c * 8;
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(abase + g * hta + (i * 2 + kb) * SUB), 16, 0, 0);
    }
}


// MI = 16-row MFMA tiles per wave along M: 8 -> the 256 x 256 tile, 10 -> a 320 x 256 tile.  The taller tile exists for wave
// quantisation: a [25600, 768] output is 300 tiles of 256^2 (two rounds on 256 CUs, the second 17 % full) but 240 tiles of
// 320 x 256 (one round); [39424, 512] is 308 against 248.  launch_nt picks the variant with the smaller rounds x tile-work.
// sum over the 16 lanes of a DPP row (= the 16 rows a lane group holds of one output column), in every lane of the row: quad swaps,
// then the half-row and row mirrors.  VALU only: no LDS round trip per step and no lane index (ds_bpermute addresses kept the lane
// id alive through the tile loop, spilled, and its reload put an s_waitcnt vmcnt(0) — every store of the tile — in front of the sums)
__device__ __forceinline__ float row16_sum(float x) {
    auto dpp = [](float v, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    x += dpp(x, std::integral_constant<int, 0xB1>{});     // quad_perm [1, 0, 3, 2]
    x += dpp(x, std::integral_constant<int, 0x4E>{});     // quad_perm [2, 3, 0, 1]
    x += dpp(x, std::integral_constant<int, 0x141>{});    // row_half_mirror
    x += dpp(x, std::integral_constant<int, 0x140>{});    // row_mirror
    return x;
}

template <int ACT, int OUT, int MI>
__global__ __launch_bounds__(512) void gemm_nt256_kernel(GemmNT p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RH = MI / 2;                      // row tiles per phase
    constexpr int AROWS = MI * 16;                  // rows of an A half-tile (one wave group's rows)
    constexpr int HTA = AROWS * BK * 2;             // bytes of an A half-tile
    constexpr int PAR = 2 * HT + 2 * HTA;           // one parity: B_lo, B_hi, A_lo, A_hi
    constexpr int BMT = 2 * AROWS;                  // tile rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // Tile raster inside an XCD's chunk: groups of group_n column tiles, rows fastest-but-one.  With n fastest over ALL column
    // tiles the 32 concurrent tiles of an XCD span every B panel (N = 3072, K = 768: 4.7 MB, more than the 4 MB L2), so B is
    // re-streamed from beyond L2 once per round; with a group whose B panels stay L2-resident an XCD streams its A rows once
    // per group and reads B once (host-side choice and traffic model: launch_nt).
    const int nwg = p.tiles_m * p.tiles_n;
    const int nk = p.K / BK;
    const int nload = 4 * nk;
    int m0, n0;
    auto locate = [&](int tl) {
        const int t = xcd_remap(tl, nwg);
        int tm, tn;
        if (p.group_n >= p.tiles_n) { tm = t / p.tiles_n; tn = t % p.tiles_n; }
        else {
            const int per = p.tiles_m * p.group_n;
            const int gi = t / per, rem = t - gi * per;
            const int left = p.tiles_n - gi * p.group_n;
            const int gw = left < p.group_n ? left : p.group_n;
            tm = rem / gw; tn = gi * p.group_n + rem % gw;
        }
        m0 = tm * BMT; n0 = tn * 256;
    };
    locate(blockIdx.x);

    f32x4 acc[MI][4];

    // half-tile l = 4*tile + w ; w: 0 B_lo, 1 B_hi, 2 A_up (upper row tiles of both wave groups), 3 A_dn (lower row tiles)
    const int ln = lane;
    auto issue = [&](int l) {
        const int tile = l >> 2, w = l & 3;
        char* par = smem + (tile & 1) * PAR;
        if (w < 2) stage_half_perm(p.B, p.ldb, n0 + w * 128, p.N, tile * BK, par + w * HT, wave, ln);
        else stage_rowhalf<RH>(p.A, p.lda, m0, p.M, tile * BK, par + 2 * HT, HTA, w - 2, wave, ln);
    };
    // profiling stamps (off unless dclip_trace_gemm_stamps armed them): s_memtime at start / first operands landed / main loop
    // done / epilogue done (stores acknowledged), s_memrealtime at start / end
    auto stamp = [&](int k) {
        if (p.stamps && tid == 0) {
            if (k == 3) WAIT_VMCNT(0);
            p.stamps[6 * (int64_t)blockIdx.x + k] = __builtin_readcyclecounter();
            if (k == 0) p.stamps[6 * (int64_t)blockIdx.x + 4] = __builtin_amdgcn_s_memrealtime();
            if (k == 3) p.stamps[6 * (int64_t)blockIdx.x + 5] = __builtin_amdgcn_s_memrealtime();
        }
    };
    stamp(0);
    if (p.clk && blockIdx.x == 0 && tid == 0) { p.clk[0] = __builtin_amdgcn_s_memtime(); p.clk[1] = __builtin_amdgcn_s_memrealtime(); }
    // LDS-DMA instructions per wave: B half 2, A row-half 2 (RH = 4) or 3 (RH = 5).  vmcnt completes in issue order, so "wait
    // until half-tile X has landed" = vmcnt(number of instructions issued after X)
#define WAIT_VM2(n8, n10) do { if (MI == 10) WAIT_VMCNT(n10); else WAIT_VMCNT(n8); } while (0)
    const int npro = nload < 7 ? nload : 7;                // tile 0 entirely, tile 1: B_lo, B_hi, A_up
    for (int l = 0; l < npro; ++l) issue(l);
    const int a_off = 2 * HT + wr * HTA;                     // this wave's A half
    const int b_off = (wc >> 1) * HT + (wc & 1) * 8 * SUB;   // this wave's B half, its 4 column blocks start at (wc&1)*4
    const int fragoff = swz((ln & 15) * 64 + (ln >> 4) * 16);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // B_lo, B_hi, A_up of K-tile 0 landed (younger: A_dn(0), B, B, A_up(1))
    if (nload > 4) WAIT_VM2(8, 10); else WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    stamp(1);
    if (wr == 1) __builtin_amdgcn_s_barrier();            // stagger: the wr = 1 group runs one barrier behind

    bf16x8 af[RH][2], b0[2][2], b1[2][2];

    // Two phases per K-tile: A = upper row tiles of the wave x all four column tiles, B = lower row tiles.  MFMA clusters of
    // 4 RH instructions between barriers, 4 barriers per K-tile.
    // Load stream: the operands of phase A of tile kt + 1 (B_lo, B_hi, A_up) are requested in phase B of tile kt - 1, A_dn(kt + 1)
    // in phase A of tile kt: every half-tile has two full phases (~2.5 k cycles) to land, which is what the loop needs when all
    // 256 CUs pull operands at once (with one phase of slack the same loop stalled on vmcnt: 3 200 instead of 2 480 cycles per
    // K-tile at 256 workgroups, while 32 workgroups ran at 2 500).  A wave waits for its own LDS reads BEFORE the phase's first
    // barrier, so whoever has passed a barrier knows that every read issued before it has completed: the regions refilled by
    // the next phase's loads are no longer being read.
    for (int kt = 0; kt < nk; ++kt) {
        const char* base = smem + (kt & 1) * PAR;
        const char* ap = base + a_off + fragoff;
        const char* bp = base + b_off + fragoff;
        // ---------------- phase A : upper row tiles ----------------
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                b0[j][kb] = *(const bf16x8*)(bp + (j * 2 + kb) * SUB);
                b1[j][kb] = *(const bf16x8*)(bp + ((2 + j) * 2 + kb) * SUB);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < RH; ++i)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) af[i][kb] = *(const bf16x8*)(ap + (i * 2 + kb) * SUB);
        if (kt + 1 < nk) { issue(4 * (kt + 1) + 3); WAIT_VM2(8, 10); }  // request A_dn(kt + 1); A_dn(kt) landed (younger: B, B, A_up, A_dn(kt + 1))
        else WAIT_VMCNT(0);
        WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < RH; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[j][kb], af[i][kb], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j][kb], af[i][kb], acc[i][2 + j], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        // ---------------- phase B : lower row tiles ----------------
#pragma unroll
        for (int i = 0; i < RH; ++i)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) af[i][kb] = *(const bf16x8*)(ap + ((RH + i) * 2 + kb) * SUB);
        // request B_lo, B_hi, A_up(kt + 2); B_lo, B_hi, A_up(kt + 1) landed (younger: A_dn(kt + 1) and the three just issued)
        if (kt + 2 < nk) { issue(4 * (kt + 2)); issue(4 * (kt + 2) + 1); issue(4 * (kt + 2) + 2); WAIT_VM2(8, 10); }
        else if (kt + 1 < nk) WAIT_VM2(2, 3);
        else WAIT_VMCNT(0);
        WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < RH; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[RH + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[j][kb], af[i][kb], acc[RH + i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[RH + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j][kb], af[i][kb], acc[RH + i][2 + j], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();            // re-align the two groups
    // (round 3 padded a dozen idle slots here against the operand re-use seen in attention_mix.hip; the static check of the shipped
    //  object — tools/asm/mfma_hazard.py, tests/test_mfma_hazard_cpu.py — shows that no VALU instruction of the epilogue writes a
    //  fragment register within 12 slots of the MFMAs that read it, and keeps showing it: the padding is gone)
    stamp(2);
    const int m0c = m0, n0c = n0;
    const int g = ln >> 4, rl = ln & 15;

    // epilogue, straight from registers (see stage_half_perm): lane (g, c) owns row c of every row tile i and, per column pair
    // jp, 8 consecutive columns.  No LDS, no barrier: a wave starts storing as soon as its own accumulators are final.  Side
    // operands (the f32 residual, which may alias C in place, and the aux rows of the DGELU / MULAUX variants) of row tile
    // i + 1 are requested before the stores of row tile i are issued, so that their in-order vmcnt wait never includes a store.
    const int colb = n0c + wc * 64 + g * 8;
    float bias[2][8], csum[2][8];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { bias[jp][e] = 0.f; csum[jp][e] = 0.f; }
        const int col = colb + jp * 32;
        if (p.bias && col < p.N) {
            const float4 q0 = *(const float4*)(p.bias + col), q1 = *(const float4*)(p.bias + col + 4);
            bias[jp][0] = q0.x; bias[jp][1] = q0.y; bias[jp][2] = q0.z; bias[jp][3] = q0.w;
            bias[jp][4] = q1.x; bias[jp][5] = q1.y; bias[jp][6] = q1.z; bias[jp][7] = q1.w;
        }
    }
    // units u = 2 i + jp, processed in batches of BU: a batch's side operands are all requested before its first store.  vmcnt
    // retires loads and stores in ONE in-order queue, so a load issued after a store cannot be waited for without also waiting
    // for that store's acknowledgement — a ring that refills one slot per unit (loads interleaved with stores) turned the
    // MULAUX epilogue into a chain of write round trips (58 k cycles per 320 x 256 tile).  With batches the wave pays that
    // once per batch boundary: none at all where the side operands of the whole tile fit the registers the dead fragments free.
    constexpr bool OUT_F32 = OUT == 1;               // (f32 units are 32 B per lane: smaller side-operand batches)
    constexpr bool SIDE = OUT != 0 || ACT == 3 || ACT == 4;
    constexpr int NU = 2 * MI;
    constexpr int BU = !SIDE ? NU : (OUT_F32 ? ((ACT == 3 || ACT == 4) ? 4 : (MI == 8 ? 8 : (MI == 6 ? 6 : 5))) : (MI == 10 ? 10 : NU));
    static_assert(NU % BU == 0, "batch size must divide the unit count");
    EpiSide side[SIDE ? BU : 1];
    const int row0 = m0c + wr * AROWS + rl;
    const int64_t o0 = (int64_t)row0 * p.ldc + colb, r0off = (int64_t)row0 * p.ldr + colb;
    const int64_t ostep = 16 * p.ldc, rstep = 16 * p.ldr;
    // full: the tile lies inside the matrix (all but the last row / column of tiles) — the per-unit bounds test is then one scalar
    // branch instead of two vector compares and an exec-mask update.  (Compiled out entirely, the 2 MI units become one basic block
    // whose addresses and conversions are all hoisted to the top: 100-240 spilled registers.)
    const bool full = m0c + BMT <= p.M && n0c + 256 <= p.N;
    auto run_units = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
        for (int ub = 0; ub < NU; ub += BU) {
            if (SIDE) {
#pragma unroll
                for (int u = ub; u < ub + BU; ++u) {
                    const int i = u >> 1, jp = u & 1;
                    if (full || (row0 + i * 16 < p.M && colb + jp * 32 < p.N))
                        epilogue_load_side<ACT, OUT>(p, o0 + i * ostep + jp * 32, r0off + i * rstep + jp * 32, side[u - ub]);
                }
            }
#pragma unroll
            for (int u = ub; u < ub + BU; ++u) {
                const int i = u >> 1, jp = u & 1;
                const int row = row0 + i * 16, col = colb + jp * 32;
                if (full || (row < p.M && col < p.N)) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] = acc[i][2 * jp][r]; v[4 + r] = acc[i][2 * jp + 1][r]; }
                    epilogue_vec8<ACT, OUT, MODE>(p, v, row, col, o0 + i * ostep + jp * 32, r0off + i * rstep + jp * 32, bias[jp], csum[jp],
                                                      side[SIDE ? u - ub : 0]);
                }
            }
        }
    };
    // (kernel arguments: the tests below are scalar, the whole workgroup takes one path)
    const bool lean = p.row_group == 0 && ((ACT == 5 || ACT == 6) ? p.aux_out != nullptr : p.aux_out == nullptr) && (OUT != 0 ? p.residual != nullptr : p.residual == nullptr);
    if constexpr (OUT != 0) {
        if (lean) run_units(std::integral_constant<int, 3>{});
        else run_units(std::integral_constant<int, 0>{});
    } else if constexpr (ACT == 3 || ACT == 4) {      // (the dgrad epilogues always come with bias-gradient column sums in the step)
        if (lean && p.colsum) run_units(std::integral_constant<int, 2>{});
        else run_units(std::integral_constant<int, 0>{});
    } else {
        if (lean && !p.colsum) run_units(std::integral_constant<int, 1>{});
        else if (lean) run_units(std::integral_constant<int, 2>{});
        else run_units(std::integral_constant<int, 0>{});
    }
    if (OUT == 0 && p.colsum) {     // (f32 / f16 output + column sums: launch_nt routes that combination to the 128^2 kernel)
        // 16-lane shuffle reduce, then the 8 waves combine through LDS (idle since the main loop's last barrier) so that the
        // workgroup issues 4 atomic wave-instructions for its 256 columns: the chip retires about one atomic wave-instruction per
        // 50 ns and CU whatever its width, and 16 four-lane atomics per wave (128 per tile) cost the MULAUX dgrad 60 us per launch
        float* cs = (float*)smem;                   // [2 wave groups][256 columns]
#pragma unroll
        for (int jp = 0; jp < 2; ++jp)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = row16_sum(csum[jp][e]);
                if (rl == 0) cs[wr * 256 + wc * 64 + jp * 32 + g * 8 + e] = x;
            }
        WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        const int tc = wave * 64 + ln;              // (= tid, from the epilogue's opaque lane index: nothing hoisted out of the tile loop)
        if (tc < 256 && n0c + tc < p.N) unsafeAtomicAdd(p.colsum + n0c + tc, cs[tc] + cs[256 + tc]);
    }
    stamp(3);
    if (p.clk && blockIdx.x == 0 && tid == 0) { p.clk[2] = __builtin_amdgcn_s_memtime(); p.clk[3] = __builtin_amdgcn_s_memrealtime(); }
}

// ------------------------------------------------------------------------------------------------------
// wgrad: out[P,Q] += sum_m A[m,P] * B[m,Q].  Both operands are "k-major" (the contraction index m is the row),
// so MFMA fragments (8 consecutive k per lane) need a transpose: tiles are staged row-major into LDS through
// registers (rows padded to 288 B) and read back with ds_read_b64_tr_b16 (guide T10).
// ------------------------------------------------------------------------------------------------------
constexpr int TP = 128, TQ = 128, TC = 64;          // output tile and contraction chunk
constexpr int TROW = 288;                           // padded LDS row stride in bytes (128 bf16 + 32 B)
constexpr int TTILE = TC * TROW;                    // 18 KiB

struct GemmTN {
    const bf16_t* A; int64_t lda;
    const bf16_t* B; int64_t ldb;
    float* out; int64_t ldo;
    int M, P, Q;
    int tiles_p, tiles_q, splits, chunk;            // chunk = rows of M per split (multiple of TC)
    unsigned long long* stamps;                     // profiling only (dclip_trace_gemm_stamps), else null
    float* partial;                                 // 256^2 wgrad: [splits][tiles][256][256] f32 partial tiles (null: f32 atomics)
};

// load a [64 x 128] bf16 tile (rows m0.., cols c0..) into registers: 4 x 16 B per thread
__device__ __forceinline__ void tn_load(const bf16_t* __restrict__ G, int64_t ld, int m0, int m_end, int c0, int cols,
                                        int tid, u32x4 (&regs)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int idx = s * 256 + tid;          // 0..1023 chunks of 16 B; 16 chunks per row
        const int r = idx >> 4, c = (idx & 15) * 8;
        const int row = m0 + r, col = c0 + c;
        if (row < m_end && col < cols) regs[s] = *(const u32x4*)(G + (int64_t)row * ld + col);
        else regs[s] = u32x4{0u, 0u, 0u, 0u};
    }
}
__device__ __forceinline__ void tn_store(char* tile, int tid, const u32x4 (&regs)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int idx = s * 256 + tid;
        const int r = idx >> 4, c = (idx & 15);
        *(u32x4*)(tile + r * TROW + c * 16) = regs[s];
    }
}

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// fragment of a k-major tile: 16 columns starting at x0, 32 rows starting at r0 (lane: col x0+(l&15), k = 8(l>>4)+e)
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int r0, int x0, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const char* a0 = tile + (r0 + 8 * g + q) * TROW + (x0 + 4 * pp) * 2;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * TROW));
    union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
    u.s.lo = lo; u.s.hi = hi;
    return u.v;
}

__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTN p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 1, wq = wave & 1;

    int t = blockIdx.x;
    const int split = t % p.splits; t /= p.splits;
    const int tq = t % p.tiles_q, tp = t / p.tiles_q;
    const int p0 = tp * TP, q0 = tq * TQ;
    const int m_begin = split * p.chunk;
    const int m_end = min(p.M, m_begin + p.chunk);
    if (m_begin >= m_end) return;

    char* As = smem;                 // [2][TTILE]
    char* Bs = smem + 2 * TTILE;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 ra[4], rb[4];
    const int nc = (m_end - m_begin + TC - 1) / TC;
    tn_load(p.A, p.lda, m_begin, m_end, p0, p.P, tid, ra);
    tn_load(p.B, p.ldb, m_begin, m_end, q0, p.Q, tid, rb);
    tn_store(As, tid, ra);
    tn_store(Bs, tid, rb);

    for (int ct = 0; ct < nc; ++ct) {
        __syncthreads();
        const bool more = ct + 1 < nc;
        if (more) {
            tn_load(p.A, p.lda, m_begin + (ct + 1) * TC, m_end, p0, p.P, tid, ra);
            tn_load(p.B, p.ldb, m_begin + (ct + 1) * TC, m_end, q0, p.Q, tid, rb);
        }
        const char* a_t = As + (ct & 1) * TTILE;
        const char* b_t = Bs + (ct & 1) * TTILE;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = tr_frag(a_t, kb * 32, wp * 64 + i * 16, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = tr_frag(b_t, kb * 32, wq * 64 + j * 16, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            // buffer (ct+1)&1 was last read in iteration ct-1; every wave passed this iteration's barrier since
            tn_store(As + ((ct + 1) & 1) * TTILE, tid, ra);
            tn_store(Bs + ((ct + 1) & 1) * TTILE, tid, rb);
        }
    }

    const int colb = q0 + wq * 64 + (lane & 15);
    const int rowb = p0 + wp * 64 + (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rowb + i * 16 + r;
            if (row >= p.P) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = colb + j * 16;
                if (col < p.Q) unsafeAtomicAdd(p.out + (int64_t)row * p.ldo + col, acc[i][j][r]);
            }
        }
}

// ------------------------------------------------------------------------------------------------------
// wgrad fast path (M % 64 == 0): same 128x128 output tile, but the k-major [64 x 128] operand tiles are staged by
// global_load_lds (16 B / lane, 4 rows per wave-instruction) into un-padded 256-byte rows with an XOR swizzle of the
// 16-byte chunk index, key(row) = 2 * ((row & 3) | ((row >> 3) & 1) << 2), applied on the source address and on the
// ds_read_b64_tr_b16 address: the 32 lanes of a half-wave (2 groups x 4 rows x 32 B) then cover all 64 banks once.
// ------------------------------------------------------------------------------------------------------
constexpr int TNB = TC * 256;                   // 16 KiB operand tile
constexpr int TCLD = 128 + 4;                   // f32 row stride of the C tile staged for the atomics
constexpr int TN_LDS = (4 * TNB > 128 * TCLD * 4) ? 4 * TNB : 128 * TCLD * 4;

__device__ __forceinline__ int tn_key(int row) { return 2 * ((row & 3) | (((row >> 3) & 1) << 2)); }

// per-lane source pointers of this wave's four pieces of a k-major [64 x 128] operand tile at the slice's first chunk (the chunk
// index adds a wave-uniform byte offset: no 64-bit multiply per piece and chunk — see tn_half_sources)
__device__ __forceinline__ void tn_glds_sources(const bf16_t* __restrict__ G, int64_t ld, int m0, int c0, int cols, int wave, int lane,
                                                const char* (&src)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int piece = wave * 4 + s;                 // 16 pieces of 4 rows
        const int r = piece * 4 + (lane >> 4);
        const int lc = (lane & 15) ^ tn_key(r);         // logical chunk that lands in physical chunk (lane & 15)
        int col = c0 + lc * 8;
        col = col < cols ? col : cols - 8;              // column edge: duplicate a valid chunk (those outputs are not stored)
        src[s] = (const char*)(G + (int64_t)(m0 + r) * ld + col);
    }
}
__device__ __forceinline__ void tn_stage_glds(const char* const (&src)[4], int64_t chunk_off, char* tile, int wave) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
        __builtin_amdgcn_global_load_lds((gbl_void*)(src[s] + chunk_off), (lds_void*)(tile + (wave * 4 + s) * 1024), 16, 0, 0);
}

__device__ __forceinline__ bf16x8 tr_frag_swz(const char* tile, int r0, int x0, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int row = r0 + 8 * g + q;
    const int chunk = (x0 >> 3) + (pp >> 1);
    const char* a0 = tile + row * 256 + ((chunk ^ tn_key(row)) << 4) + (pp & 1) * 8;
    union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
    u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * 256));      // rows + 4: same key
    return u.v;
}

// The same fragment through inline asm, for gemm_tn256_kernel: hipcc puts `s_waitcnt vmcnt(0)` in front of every
// __builtin_amdgcn_ds_read_tr16_b64 that follows an LDS-DMA in program order (it cannot tell that the read does not alias the
// in-flight destination), which drained the six-half-tile load stream of that kernel TWICE per chunk — the counted vmcnt waits
// below were dead letters.  An asm read is invisible to that pass; its completion is covered by the explicit lgkmcnt(0) before
// each phase's barrier, fenced against the MFMAs by a sched_barrier (cdna_hip_programming.md section 5.4 rule 18).
// base = LDS byte address of the fragment at contraction offset 0; OFF = rows * 256 (rows + 32 keep the swizzle key).
template <int OFF>
__device__ __forceinline__ bf16x8 tr_frag_asm(unsigned base) {
    union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u.s.lo) : "v"(base), "n"(OFF));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u.s.hi) : "v"(base), "n"(OFF + 4 * 256));
    return u.v;
}
__device__ __forceinline__ unsigned tr_frag_base(const char* tile, int x0, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int row = 8 * g + q;
    const int chunk = (x0 >> 3) + (pp >> 1);
    return (unsigned)(uintptr_t)(tile + row * 256 + ((chunk ^ tn_key(row)) << 4) + (pp & 1) * 8);
}

__global__ __launch_bounds__(256) void gemm_tn_glds_kernel(GemmTN p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 1, wq = wave & 1;

    // split-major order cut into one chunk per XCD, as in gemm_tn256_kernel: co-resident workgroups of an XCD share operand slices
    const int tiles = p.tiles_p * p.tiles_q;
    const int t = xcd_remap(blockIdx.x, tiles * p.splits);
    const int split = t / tiles, tile = t - split * tiles;
    const int tq = tile % p.tiles_q, tp = tile / p.tiles_q;
    const int p0 = tp * TP, q0 = tq * TQ;
    const int m_begin = split * p.chunk;
    const int m_end = min(p.M, m_begin + p.chunk);
    if (m_begin >= m_end) return;
    const int nc = (m_end - m_begin) / TC;              // M % 64 == 0 and chunk % 64 == 0

    char* As = smem;
    char* Bs = smem + 2 * TNB;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const char* src_a[4];
    const char* src_b[4];
    tn_glds_sources(p.A, p.lda, m_begin, p0, p.P, wave, lane, src_a);
    tn_glds_sources(p.B, p.ldb, m_begin, q0, p.Q, wave, lane, src_b);
    const int64_t step_a = (int64_t)TC * p.lda * 2, step_b = (int64_t)TC * p.ldb * 2;        // bytes per chunk (wave-uniform)
    tn_stage_glds(src_a, 0, As, wave);
    tn_stage_glds(src_b, 0, Bs, wave);
    for (int ct = 0; ct < nc; ++ct) {
        __syncthreads();
        if (ct + 1 < nc) {
            const int nb = (ct + 1) & 1;
            tn_stage_glds(src_a, (ct + 1) * step_a, As + nb * TNB, wave);
            tn_stage_glds(src_b, (ct + 1) * step_b, Bs + nb * TNB, wave);
        }
        const char* a_t = As + (ct & 1) * TNB;
        const char* b_t = Bs + (ct & 1) * TNB;
        // (asm reads: behind the builtin form hipcc waits vmcnt(0) for the LDS-DMA of chunk ct + 1 just issued above, i.e. the
        //  prefetch was drained before the chunk it was meant to overlap: tr_frag_asm)
        unsigned abase[4], bbase[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { abase[i] = tr_frag_base(a_t, wp * 64 + i * 16, lane); bbase[i] = tr_frag_base(b_t, wq * 64 + i * 16, lane); }
        bf16x8 af[2][4], bfr[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { af[0][i] = tr_frag_asm<0>(abase[i]); bfr[0][i] = tr_frag_asm<0>(bbase[i]); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { af[1][i] = tr_frag_asm<32 * 256>(abase[i]); bfr[1][i] = tr_frag_asm<32 * 256>(bbase[i]); }
        WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kb][i], bfr[kb][j], acc[i][j], 0, 0, 0);
    }
    // epilogue: stage the 128x128 f32 tile through LDS (operand buffers are dead) so that every atomic wave-instruction
    // adds 64 consecutive floats of one output row = 256 contiguous bytes (the full-rate shape, MI355X_MICROARCH.md
    // "Global float atomics"), instead of four 64-byte segments straight from the accumulator layout.
    __syncthreads();
    float* cs = (float*)smem;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                cs[(wp * 64 + i * 16 + (lane >> 4) * 4 + r) * TCLD + wq * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
    __syncthreads();
#pragma unroll 4
    for (int it = 0; it < 64; ++it) {
        const int rl = wave * 32 + (it >> 1), cl = (it & 1) * 64 + lane;
        const int row = p0 + rl, col = q0 + cl;
        if (row < p.P && col < p.Q) unsafeAtomicAdd(p.out + (int64_t)row * p.ldo + col, cs[rl * TCLD + cl]);
    }
}

// ------------------------------------------------------------------------------------------------------
// wgrad on the 256x256 staggered pipeline (P % 256 == 0, Q % 256 == 0, M % 64 == 0): same half-tile ring, phase schedule,
// counted vmcnt and wave-group stagger as gemm_nt256_kernel; the half-tiles are k-major [64 rows of M][128 columns]
// (256-byte rows, tn_key swizzle), fragments come from ds_read_b64_tr_b16.  Every workgroup owns one output tile and one
// slice of the token axis (long contraction = the regime this pipeline is best at); partial tiles are added with
// row-contiguous f32 atomics staged through LDS.
// ------------------------------------------------------------------------------------------------------
// per-lane source pointers of this wave's two LDS-DMA pieces of a k-major half-tile at the slice's first chunk: the chunk index only
// adds a wave-uniform byte offset.  (Formed per issue as G + (m0 + r) * ld + c, the compiler re-did the 64-bit multiply for every
// piece: 16 v_mul_lo_u32 + 8 v_mad_u64_u32 — quarter-rate instructions — per wave and chunk, ~1 500 issue cycles per SIMD and chunk
// next to 2 048 cycles of MFMA: the contraction loop ran 3 600 cycles per chunk where gemm_nt256_kernel's runs 2 465.)
__device__ __forceinline__ void tn_half_sources(const bf16_t* __restrict__ G, int64_t ld, int m0, int c0, int wave, int lane,
                                                const char* (&src)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int piece = wave * 2 + s;                 // 16 pieces of 4 rows per half-tile, 2 per wave
        const int r = piece * 4 + (lane >> 4);
        const int lc = (lane & 15) ^ tn_key(r);
        src[s] = (const char*)(G + (int64_t)(m0 + r) * ld + c0 + lc * 8);
    }
}
__device__ __forceinline__ void stage_half_tn(const char* const (&src)[2], int64_t chunk_off, char* buf, int wave) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
        __builtin_amdgcn_global_load_lds((gbl_void*)(src[s] + chunk_off), (lds_void*)(buf + (wave * 2 + s) * 1024), 16, 0, 0);
}

__global__ __launch_bounds__(512) void gemm_tn256_kernel(GemmTN p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // Workgroup -> (split, tile): all workgroups are resident at once (~one per CU), so what matters is WHICH of them share an
    // XCD's L2.  The tiles of one split read the same token rows: tile (tp, tq) needs the A panel slice tp and the B panel slice
    // tq of that split.  Linear order split-major / tq fastest, cut into one contiguous chunk per XCD (xcd_remap): an XCD's ~32
    // workgroups are then (almost) all tiles of ONE split — 12 + 3 panel slices for 32 workgroups instead of two private
    // slices each (the split-fastest order of round 1 fetched 1.62 GB per launch against 0.43 GB algorithmic).
    const int tiles = p.tiles_p * p.tiles_q;
    const int t = xcd_remap(blockIdx.x, tiles * p.splits);
    const int split = t / tiles, tile = t - split * tiles;
    const int tq = tile % p.tiles_q, tp = tile / p.tiles_q;
    const int p0 = tp * 256, q0 = tq * 256;
    const int m_begin = split * p.chunk;
    const int m_end = min(p.M, m_begin + p.chunk);
    if (m_begin >= m_end) return;
    const int nk = (m_end - m_begin) / TC;
    const int nload = 4 * nk;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // half-tile l = 4*chunk + w ; w: 0 B_lo, 1 B_hi (X columns q0 + w*128), 2 A_lo, 3 A_hi (dY columns p0 + (w-2)*128)
    const char* srcs[4][2];
    tn_half_sources(p.B, p.ldb, m_begin, q0, wave, lane, srcs[0]);
    tn_half_sources(p.B, p.ldb, m_begin, q0 + 128, wave, lane, srcs[1]);
    tn_half_sources(p.A, p.lda, m_begin, p0, wave, lane, srcs[2]);
    tn_half_sources(p.A, p.lda, m_begin, p0 + 128, wave, lane, srcs[3]);
    const int64_t step_b = (int64_t)TC * p.ldb * 2, step_a = (int64_t)TC * p.lda * 2;      // bytes per chunk (wave-uniform)
    auto issue = [&](int l) {
        const int kt = l >> 2, w = l & 3;
        char* buf = smem + ((kt & 1) * 4 + w) * HT;
        // (w is a compile-time constant at every call site once the loop body is unrolled: l = 4 * chunk + const)
        if (w == 0) stage_half_tn(srcs[0], kt * step_b, buf, wave);
        else if (w == 1) stage_half_tn(srcs[1], kt * step_b, buf, wave);
        else if (w == 2) stage_half_tn(srcs[2], kt * step_a, buf, wave);
        else stage_half_tn(srcs[3], kt * step_a, buf, wave);
    };
    auto stamp = [&](int k) {
        if (p.stamps && tid == 0) {
            if (k == 3) WAIT_VMCNT(0);
            p.stamps[6 * (int64_t)blockIdx.x + k] = __builtin_readcyclecounter();
            if (k == 0) p.stamps[6 * (int64_t)blockIdx.x + 4] = __builtin_amdgcn_s_memrealtime();
            if (k == 3) p.stamps[6 * (int64_t)blockIdx.x + 5] = __builtin_amdgcn_s_memrealtime();
        }
    };
    stamp(0);
    // Two phases per contraction chunk (rows 0-63 / 64-127 of the wave's 128 output rows x its 64 columns), as in
    // gemm_nt256_kernel: 4 barriers per chunk, MFMA clusters of 32.  Loads run 6 half-tiles ahead (the A halves of chunk kt + 1
    // are requested in phase A of chunk kt, the B halves of chunk kt + 2 in phase B); a wave retires its own LDS reads before
    // the phase's first barrier, so a region is only refilled once nobody reads it any more.
    const int npro = nload < 6 ? nload : 6;
#pragma unroll
    for (int l = 0; l < 6; ++l)
        if (l < npro) issue(l);          // (constant l: the source table stays in registers)
    if (nload > 4) WAIT_VMCNT(4); else WAIT_VMCNT(0);      // (two B halves of chunk 1 stay in flight: 2 instructions each)
    __builtin_amdgcn_s_barrier();
    stamp(1);
    if (wr == 1) __builtin_amdgcn_s_barrier();

    const int a_off = (2 + wr) * HT;
    const int b_off = (wc >> 1) * HT;
    const int bcol = (wc & 1) * 64;                    // this wave's 64 columns inside its B half
    bf16x8 af[4][2], b0[2][2], b1[2][2];

    for (int kt = 0; kt < nk; ++kt) {
        const char* base = smem + (kt & 1) * 4 * HT;
        const char* at = base + a_off;
        const char* bt = base + b_off;
        const int k4 = kt * 4;
        // phase A : rows 0-63
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned a0 = tr_frag_base(bt, bcol + j * 16, lane), a1 = tr_frag_base(bt, bcol + (2 + j) * 16, lane);
            b0[j][0] = tr_frag_asm<0>(a0); b0[j][1] = tr_frag_asm<32 * 256>(a0);
            b1[j][0] = tr_frag_asm<0>(a1); b1[j][1] = tr_frag_asm<32 * 256>(a1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned a0 = tr_frag_base(at, i * 16, lane);
            af[i][0] = tr_frag_asm<0>(a0); af[i][1] = tr_frag_asm<32 * 256>(a0);
        }
        if (k4 + 6 < nload) issue(k4 + 6);
        if (k4 + 7 < nload) issue(k4 + 7);
        WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][kb], b0[j][kb], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][kb], b1[j][kb], acc[i][2 + j], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        // phase B : rows 64-127
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned a0 = tr_frag_base(at, (4 + i) * 16, lane);
            af[i][0] = tr_frag_asm<0>(a0); af[i][1] = tr_frag_asm<32 * 256>(a0);
        }
        if (k4 + 8 < nload) { issue(k4 + 8); issue(k4 + 9); WAIT_VMCNT(4); }       // chunk kt + 1 landed
        else WAIT_VMCNT(0);
        WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][kb], b0[j][kb], acc[4 + i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[4 + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][kb], b1[j][kb], acc[4 + i][2 + j], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    // (no idle slots before the epilogue: see gemm_nt256_kernel)
    stamp(2);

    // epilogue: 4 slabs of 64 rows through LDS, then row-contiguous f32 atomics (256 lanes x 4 B = 1 KiB per row)
    float* cs = (float*)smem;
    constexpr int CL = 256 + 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    cs[(wr * 32 + i * 16 + (lane >> 4) * 4 + r) * CL + wc * 64 + j * 16 + (lane & 15)] = acc[q * 2 + i][j][r];
        __syncthreads();
        if (p.partial) {
            // partial tile with plain 16-byte stores (the chip adds f32 atomics at ~1.3 TB/s, stores run at ~6): the splits of a
            // tile are summed into dW by tn256_reduce_kernel afterwards, in a fixed order -> run-to-run identical gradients
            float* dst = p.partial + ((int64_t)split * tiles + tile) * 65536;
#pragma unroll 4
            for (int it = 0; it < 8; ++it) {
                const int sl = (tid >> 6) + it * 8;                      // slab row 0..63, 64 threads x float4 per row
                const int rt = (sl >> 5) * 128 + q * 32 + (sl & 31);     // row inside the tile
                *(float4*)(dst + rt * 256 + (tid & 63) * 4) = *(const float4*)&cs[sl * CL + (tid & 63) * 4];
            }
        } else {
#pragma unroll 4
            for (int it = 0; it < 32; ++it) {
                const int sl = (tid >> 8) + it * 2;                          // slab row 0..63, 256 threads per row
                const int row = p0 + (sl >> 5) * 128 + q * 32 + (sl & 31);
                const int col = q0 + (tid & 255);
                unsafeAtomicAdd(p.out + (int64_t)row * p.ldo + col, cs[sl * CL + (tid & 255)]);
            }
        }
    }
    stamp(3);
}

// dW[row, col .. col + 3] += sum over the splits of the partial tiles written by gemm_tn256_kernel (fixed order)
__global__ __launch_bounds__(256) void tn256_reduce_kernel(const float* __restrict__ partial, int splits, int tiles, int tiles_q,
                                                           float* __restrict__ out, int64_t ldo) {
    const int idx = blockIdx.x * 256 + threadIdx.x;          // float4 index: tile, row in tile, column quad
    const int tile = idx >> 14, rt = (idx >> 6) & 255, c4 = idx & 63;
    if (tile >= tiles) return;
    const float* src = partial + (int64_t)tile * 65536 + rt * 256 + c4 * 4;
    float4 a = *(const float4*)src;
    for (int s = 1; s < splits; ++s) {
        const float4 v = *(const float4*)(src + (int64_t)s * tiles * 65536);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    float* o = out + (int64_t)((tile / tiles_q) * 256 + rt) * ldo + (tile % tiles_q) * 256 + c4 * 4;
    float4 cur = *(const float4*)o;
    cur.x += a.x; cur.y += a.y; cur.z += a.z; cur.w += a.w;
    *(float4*)o = cur;
}

// column sums: grid (ceil(N/256) , row_splits); each thread owns one column, strides rows
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ X, int64_t ld, float* __restrict__ db,
                                                     int M, int N, int rows_per_block) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= N) return;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
        s0 += bf2f(X[(int64_t)r * ld + col]);
        s1 += bf2f(X[(int64_t)(r + 1) * ld + col]);
        s2 += bf2f(X[(int64_t)(r + 2) * ld + col]);
        s3 += bf2f(X[(int64_t)(r + 3) * ld + col]);
    }
    for (; r < r1; ++r) s0 += bf2f(X[(int64_t)r * ld + col]);
    unsafeAtomicAdd(db + col, (s0 + s1) + (s2 + s3));
}

// 16-byte form: thread (cg = tid & 31, rl = tid >> 5) sums 8 consecutive columns over rows r0 + rl, r0 + rl + 8, ... ; the 8 row
// lanes of a column group are combined through LDS, one atomic per column and block
__global__ __launch_bounds__(256) void colsum8_kernel(const bf16_t* __restrict__ X, int64_t ld, float* __restrict__ db,
                                                      int M, int N, int rows_per_block) {
    __shared__ float red[8][256 + 8];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int col = blockIdx.x * 256 + cg * 8;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (col < N) {
        int r = r0 + rl;
        for (; r + 24 < r1; r += 32) {                       // four rows in flight per thread
            bf16x8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const bf16x8*)(X + (int64_t)(r + 8 * u) * ld + col);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[u][e]);
        }
        for (; r < r1; r += 8) {
            const bf16x8 v = *(const bf16x8*)(X + (int64_t)r * ld + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][cg * 8 + e] = acc[e];
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < N) {
        float x = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) x += red[w][threadIdx.x];
        unsafeAtomicAdd(db + c, x);
    }
}

// DCLIP_GEMM256=0: every shape on the 128^2 kernel (the one fallback of this stage); default: the 256- / 320- / 192-row kernel for
// every shape it tiles — with the towers on four streams, other kernels fill its tail rounds
bool use_256(const GemmNT& p) {
    static const int mode = [] { const char* e = getenv("DCLIP_GEMM256"); return e ? atoi(e) : 1; }();
    return mode != 0 && p.M >= 1024 && p.N >= 256;
}

// clock probe (dclip_trace_gemm_clock): slot of the next 256- / 320- / 192-row launch, null when the probe is off
std::atomic<unsigned long long*> g_clock_buf{nullptr};
std::atomic<long long> g_clock_cap{0};
std::atomic<long long> g_clock_count{0};
inline unsigned long long* next_clock_slot() {
    unsigned long long* buf = g_clock_buf.load(std::memory_order_acquire);
    const long long cap = g_clock_cap.load(std::memory_order_relaxed);
    if (!buf || cap <= 0) return nullptr;
    return buf + 4 * (g_clock_count.fetch_add(1, std::memory_order_relaxed) % cap);
}

template <int ACT>
int launch_nt(GemmNT p, int out, hipStream_t st) {
    // OUT = 2 (f16 output + f16 in-place residual: the frozen teacher's residual stream) exists for the plain epilogue only
    if constexpr (ACT != 0) { if (out == 2) { dclip_set_error("dclip_gemm_nt: f16 output needs act = DCLIP_ACT_NONE"); return DCLIP_EINVAL; } }
    const bool out_f32 = out == 1;
    // (the 256- / 320-row kernels keep column sums only with bf16 output, and an activation only with bf16 output: the step's activated
    //  GEMMs all store bf16 — an f32 store after an activation is served by the 128^2 kernel, 15 instantiations less in the library)
    if (use_256(p) && !(out != 0 && p.colsum) && (ACT == 0 || out == 0)) {
        int tm = (p.M + 255) / 256;
        const int tn = (p.N + 255) / 256;
        // tile height: 192, 256 or 320 rows (MI = 6, 8, 10), whichever needs the fewest rounds x cycles per tile on 256 CUs (a tie keeps
        // the 256-row tile, whose loads carry no duplicates); DCLIP_GEMM320=0 keeps 256 rows, =2 forces 320, =6 forces 192
        static const int mode320 = [] { const char* e = getenv("DCLIP_GEMM320"); return e ? atoi(e) : 1; }();
        int mi = 8;
        // (every epilogue variant fits the 160 accumulator registers of the 320-row tile without spilling: tools/diag/regs.py)
        if (mode320) {
            // rounds on 256 CUs x cycles per tile, from the in-kernel stamps (tools/diag/gemm_phases.py): prologue 3.3 k, main
            // loop 2 750 (256 rows) / 3 200 (320 rows) cycles per k-tile — the taller tile does 1.25 x the work in 1.16 x the
            // time; the 192-row tile (M = 12 800 problems, where 320 x 256 tiles leave half the CUs idle at N = 768) is priced at
            // 0.78 x — and an epilogue that scales with the rows (bf16 ~9.5 k, f32 + residual ~40 k when the whole chip stores)
            const double nkt = (double)(p.K / BK), e8 = out_f32 ? 40000.0 : out == 2 ? 24000.0 : (ACT == 0 || ACT == 4 ? 9500.0 : 14000.0);
            auto cost = [&](int rows, double per_kt, double escale) {
                const int tmx = (p.M + rows - 1) / rows;
                return (double)((tmx * tn + 255) / 256) * (3300.0 + nkt * per_kt + escale * e8);
            };
            const double c8 = cost(256, 2750.0, 1.0), c10 = cost(320, 3200.0, 1.25), c6 = cost(192, 2150.0, 0.75);
            if (mode320 == 2) mi = 10;
            else if (mode320 == 6) mi = 6;
            else {
                if (c10 < c8) mi = 10;
                if (c6 < (mi == 10 ? c10 : c8) * 0.97) mi = 6;
            }
            tm = (p.M + mi * 32 - 1) / (mi * 32);
        }
        p.tiles_m = tm; p.tiles_n = tn;
        p.clk = next_clock_slot();
        const int ntiles = p.tiles_m * p.tiles_n;
        // raster group width: minimise the modelled operand bytes from beyond L2 —  A once per group, B once per XCD while a
        // group's B panels (256 x K bf16 each) fit ~2.5 MB of the XCD's 4 MB L2, else once per round of 32 tiles per XCD
        {
            const double panel = 512.0 * (double)p.K, a_bytes = 2.0 * (double)p.M * (double)p.K;
            double best = 0.0; int best_g = tn;
            for (int g = 1; g <= tn; ++g) {
                const int ngroups = (tn + g - 1) / g;
                const double b_term = (g * panel <= 2.5e6) ? 8.0 * g * panel * (ngroups > 8 ? ngroups / 8.0 : 1.0)
                                                           : (double)ntiles / 32.0 * g * panel;
                const double cost = a_bytes * ngroups + b_term;
                if (g == 1 || cost < best * 0.999) { best = cost; best_g = g; }
            }
            p.group_n = best_g;
        }
        if (mi == 10) {
            const size_t lds320 = 2 * (2 * HT + 2 * 160 * BK * 2) + 16;
            if (out_f32) { if constexpr (ACT == 0) hipLaunchKernelGGL((gemm_nt256_kernel<0, 1, 10>), dim3(ntiles), dim3(512), lds320, st, p); }
            else if (out == 2) { if constexpr (ACT == 0) hipLaunchKernelGGL((gemm_nt256_kernel<0, 2, 10>), dim3(ntiles), dim3(512), lds320, st, p); }
            else hipLaunchKernelGGL((gemm_nt256_kernel<ACT, 0, 10>), dim3(ntiles), dim3(512), lds320, st, p);
            return dclip_check_launch("dclip_gemm_nt");
        }
        if (mi == 6) {
            const size_t lds192 = 2 * (2 * HT + 2 * 96 * BK * 2) + 16;
            if (out_f32) { if constexpr (ACT == 0) hipLaunchKernelGGL((gemm_nt256_kernel<0, 1, 6>), dim3(ntiles), dim3(512), lds192, st, p); }
            else if (out == 2) { if constexpr (ACT == 0) hipLaunchKernelGGL((gemm_nt256_kernel<0, 2, 6>), dim3(ntiles), dim3(512), lds192, st, p); }
            else hipLaunchKernelGGL((gemm_nt256_kernel<ACT, 0, 6>), dim3(ntiles), dim3(512), lds192, st, p);
            return dclip_check_launch("dclip_gemm_nt");
        }
        const size_t lds256 = 8 * HT + 16;
        if (out_f32) { if constexpr (ACT == 0) hipLaunchKernelGGL((gemm_nt256_kernel<0, 1, 8>), dim3(ntiles), dim3(512), lds256, st, p); }
        else if (out == 2) { if constexpr (ACT == 0) hipLaunchKernelGGL((gemm_nt256_kernel<0, 2, 8>), dim3(ntiles), dim3(512), lds256, st, p); }
        else hipLaunchKernelGGL((gemm_nt256_kernel<ACT, 0, 8>), dim3(ntiles), dim3(512), lds256, st, p);
        return dclip_check_launch("dclip_gemm_nt");
    }
    const int grid = p.tiles_m * p.tiles_n;
    const size_t lds = NT_LDS;
    if (out_f32) hipLaunchKernelGGL((gemm_nt_kernel<ACT, 1>), dim3(grid), dim3(256), lds, st, p);
    else if (out == 2) { if constexpr (ACT == 0) hipLaunchKernelGGL((gemm_nt_kernel<0, 2>), dim3(grid), dim3(256), lds, st, p); }
    else hipLaunchKernelGGL((gemm_nt_kernel<ACT, 0>), dim3(grid), dim3(256), lds, st, p);
    return dclip_check_launch("dclip_gemm_nt");
}

unsigned long long* g_gemm_stamps = nullptr;
std::atomic<long long> g_tn_atomic_fallbacks{0};

}  // namespace

extern "C" int dclip_trace_gemm_stamps(void* buf) { g_gemm_stamps = (unsigned long long*)buf; return 0; }
extern "C" int64_t dclip_trace_gemm_clock(void* buf, int64_t cap) {
    const long long n = g_clock_count.exchange(0, std::memory_order_relaxed);
    g_clock_cap.store(buf ? (long long)cap : 0, std::memory_order_relaxed);
    g_clock_buf.store((unsigned long long*)buf, std::memory_order_release);
    return (int64_t)n;
}
extern "C" int64_t dclip_gemm_tn_atomic_fallbacks(void) { return (int64_t)g_tn_atomic_fallbacks.load(std::memory_order_relaxed); }

extern "C" int dclip_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                             int64_t M, int64_t N, int64_t K, float alpha, const float* bias, int act,
                             const void* aux_in, void* aux_out, const void* residual, int64_t ldr, int out_dtype,
                             int64_t row_group, const float* rowadd, float* colsum_acc, void* stream) {
    DCLIP_REQUIRE(A && B && C, "dclip_gemm_nt: null operand");
    DCLIP_REQUIRE(M > 0 && N > 0 && K > 0, "dclip_gemm_nt: empty problem M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
    DCLIP_REQUIRE(K % BK == 0, "dclip_gemm_nt: K=%ld must be a multiple of %d", (long)K, BK);
    DCLIP_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0,
                  "dclip_gemm_nt: operand rows must be 16-byte aligned (lda=%ld ldb=%ld)", (long)lda, (long)ldb);
    DCLIP_REQUIRE(N % 8 == 0 && ldc % 8 == 0 && ((uintptr_t)C % 16) == 0, "dclip_gemm_nt: N and ldc must be multiples of 8 and C 16-byte aligned (N=%ld ldc=%ld)", (long)N, (long)ldc);
    DCLIP_REQUIRE(out_dtype >= 0 && out_dtype <= 2, "dclip_gemm_nt: bad output dtype %d (0 bf16, 1 f32, 2 f16)", out_dtype);
    DCLIP_REQUIRE(!residual || (ldr % (out_dtype == 2 ? 8 : 4) == 0 && ((uintptr_t)residual % 16) == 0), "dclip_gemm_nt: residual rows must be 16-byte aligned");
    DCLIP_REQUIRE(act >= 0 && act <= 6, "dclip_gemm_nt: bad activation code %d", act);
    DCLIP_REQUIRE(out_dtype != 2 || act == 0, "dclip_gemm_nt: f16 output needs act = DCLIP_ACT_NONE");
    DCLIP_REQUIRE((act != DCLIP_ACT_DGELU && act != DCLIP_ACT_MULAUX) || aux_in, "dclip_gemm_nt: DGELU / MULAUX need aux_in");
    DCLIP_REQUIRE(!aux_in || ((uintptr_t)aux_in % 16) == 0, "dclip_gemm_nt: aux_in must be 16-byte aligned");
    DCLIP_REQUIRE(!aux_out || ((uintptr_t)aux_out % 16) == 0, "dclip_gemm_nt: aux_out must be 16-byte aligned");
    DCLIP_REQUIRE(row_group == 0 || rowadd, "dclip_gemm_nt: row_group needs rowadd");
    DCLIP_REQUIRE(M < (1LL << 31) && N < (1LL << 31), "dclip_gemm_nt: dimension overflow");
    GemmNT p;
    p.A = (const bf16_t*)A; p.lda = lda; p.B = (const bf16_t*)B; p.ldb = ldb; p.C = C; p.ldc = ldc;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.alpha = alpha; p.bias = bias;
    p.aux_in = aux_in; p.aux_out = aux_out; p.residual = residual; p.ldr = ldr;
    p.row_group = (int)row_group; p.rowadd = rowadd; p.colsum = colsum_acc;
    p.tiles_m = (int)((M + BM - 1) / BM); p.tiles_n = (int)((N + BN - 1) / BN);
    p.stamps = g_gemm_stamps;
    p.group_n = 1 << 30;
    p.clk = nullptr;
    hipStream_t st = (hipStream_t)stream;
    // algorithmic bytes of the call: both operands once, the output once, plus what the fused epilogue consumes / produces — the f32 residual
    // it adds (read), the saved pre-activation / derivative it multiplies by (aux_in) or stores (aux_out); round 3 counted operands + output only
    const int out_f32 = out_dtype == 1;
    const double aux_b = (act == DCLIP_ACT_MULAUX || act == DCLIP_ACT_GELU_SAVE || act == DCLIP_ACT_QUICKGELU_SAVE) ? 1.0 : 2.0;      // the saved gelu' is one byte per element
    TraceScope tr(DCLIP_TRACE_GEMM_NT, 2.0 * (double)M * (double)N * (double)K,
                  2.0 * ((double)M * K + (double)N * K) + ((out_f32 ? 4.0 : 2.0) + (residual ? (out_dtype == 2 ? 2.0 : 4.0) : 0.0) + (aux_in ? aux_b : 0.0) + (aux_out ? aux_b : 0.0)) * (double)M * N,
                  stream, (int)M, (int)N, (int)K,
                  act + 8 * (out_f32 != 0) + 16 * (residual != nullptr) + 32 * (colsum_acc != nullptr) + 64 * (out_dtype == 2));
    switch (act) {
        case 0: return launch_nt<0>(p, out_dtype, st);
        case 1: return launch_nt<1>(p, out_dtype, st);
        case 2: return launch_nt<2>(p, out_dtype, st);
        case 3: return launch_nt<3>(p, out_dtype, st);
        case 4: return launch_nt<4>(p, out_dtype, st);
        case 5: return launch_nt<5>(p, out_dtype, st);
        default: return launch_nt<6>(p, out_dtype, st);
    }
}

// room for the partial tiles of the largest 256^2 wgrad launch (~one workgroup per CU, plus rounding)
extern "C" size_t dclip_gemm_tn_workspace_bytes(void) { return (size_t)384 * 65536 * sizeof(float); }

extern "C" int dclip_gemm_tn_acc(const void* A, int64_t lda, const void* B, int64_t ldb, float* dW, int64_t ldo,
                                 int64_t M, int64_t P, int64_t Q, int splits, void* workspace, size_t ws_bytes, void* stream) {
    DCLIP_REQUIRE(A && B && dW, "dclip_gemm_tn_acc: null operand");
    DCLIP_REQUIRE(M > 0 && P > 0 && Q > 0, "dclip_gemm_tn_acc: empty problem");
    DCLIP_REQUIRE(P % 8 == 0 && Q % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 &&
                  ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0,
                  "dclip_gemm_tn_acc: P, Q, lda, ldb must be multiples of 8 and bases 16-byte aligned");
    DCLIP_REQUIRE(splits >= 1, "dclip_gemm_tn_acc: splits must be >= 1");
    GemmTN p;
    p.A = (const bf16_t*)A; p.lda = lda; p.B = (const bf16_t*)B; p.ldb = ldb; p.out = dW; p.ldo = ldo;
    p.M = (int)M; p.P = (int)P; p.Q = (int)Q; p.stamps = g_gemm_stamps; p.partial = nullptr;
    p.tiles_p = (int)((P + TP - 1) / TP); p.tiles_q = (int)((Q + TQ - 1) / TQ);
    int chunk = (int)((M + splits - 1) / splits);
    chunk = ((chunk + TC - 1) / TC) * TC;
    p.chunk = chunk;
    p.splits = (int)((M + chunk - 1) / chunk);
    const int grid = p.tiles_p * p.tiles_q * p.splits;
    const bool fast = M % TC == 0 && P >= 8 && Q >= 8;
    // 256^2 pipeline for the large outputs: one workgroup per CU, every workgroup a long slice of the token axis
    if (fast && P % 256 == 0 && Q % 256 == 0 && (P / 256) * (Q / 256) >= 16 && M >= 4096) {
        GemmTN q = p;
        q.tiles_p = (int)(P / 256); q.tiles_q = (int)(Q / 256);
        const int tiles = q.tiles_p * q.tiles_q;
        int s = (256 + tiles / 2) / tiles;                      // ~ one workgroup per CU
        if (s < 1) s = 1;
        int ch = (int)((M + s - 1) / s);
        ch = ((ch + TC - 1) / TC) * TC;
        q.chunk = ch;
        q.splits = (int)((M + ch - 1) / ch);
        TraceScope tr2(DCLIP_TRACE_GEMM_TN, 2.0 * (double)M * (double)P * (double)Q, 2.0 * ((double)M * P + (double)M * Q) + 4.0 * (double)P * Q, stream, (int)M, (int)P, (int)Q, 256);
        static const int tn_partial = [] { const char* e = getenv("DCLIP_TN_PARTIAL"); return e ? atoi(e) : 1; }();
        const size_t need = (size_t)tiles * q.splits * 65536 * sizeof(float);
        const bool part = tn_partial != 0 && workspace && ws_bytes >= need && ldo % 4 == 0 && ((uintptr_t)dW % 16) == 0 && q.splits > 1;
        // (diagnostic: the f32-atomic path is correct but not run-to-run identical; callers that rely on bit-reproducible wgrads
        //  check that this counter stays at zero)
        if (!part && q.splits > 1) g_tn_atomic_fallbacks.fetch_add(1, std::memory_order_relaxed);
        q.partial = part ? (float*)workspace : nullptr;
        hipLaunchKernelGGL(gemm_tn256_kernel, dim3(tiles * q.splits), dim3(512), 8 * HT, (hipStream_t)stream, q);
        if (part)
            hipLaunchKernelGGL(tn256_reduce_kernel, dim3((unsigned)(tiles * 64)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace,
                               q.splits, tiles, q.tiles_q, dW, ldo);
        return dclip_check_launch("dclip_gemm_tn_acc");
    }
    TraceScope tr(DCLIP_TRACE_GEMM_TN, 2.0 * (double)M * (double)P * (double)Q, 2.0 * ((double)M * P + (double)M * Q) + 4.0 * (double)P * Q, stream, (int)M, (int)P, (int)Q, 128);
    if (fast) hipLaunchKernelGGL(gemm_tn_glds_kernel, dim3(grid), dim3(256), TN_LDS, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(gemm_tn_kernel, dim3(grid), dim3(256), 4 * TTILE, (hipStream_t)stream, p);
    return dclip_check_launch("dclip_gemm_tn_acc");
}

extern "C" int dclip_colsum_acc(const void* X, int64_t ld, float* db, int64_t M, int64_t N, void* stream) {
    DCLIP_REQUIRE(X && db && M > 0 && N > 0, "dclip_colsum_acc: bad argument");
    // enough row slices to put >= ~512 workgroups on the chip (a [512, 512] head gradient used to run on 2 workgroups)
    const int colblocks = (int)((N + 255) / 256);
    int rows_per_block = (int)((M * colblocks + 511) / 512);
    rows_per_block = rows_per_block < 16 ? 16 : (rows_per_block > 512 ? 512 : (rows_per_block + 3) & ~3);
    dim3 grid((unsigned)((N + 255) / 256), (unsigned)((M + rows_per_block - 1) / rows_per_block));
    if (N % 8 == 0 && ld % 8 == 0 && ((uintptr_t)X & 15) == 0)
        hipLaunchKernelGGL(colsum8_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)X, ld, db, (int)M, (int)N,
                           rows_per_block);
    else
        hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)X, ld, db, (int)M, (int)N,
                           rows_per_block);
    return dclip_check_launch("dclip_colsum_acc");
}
