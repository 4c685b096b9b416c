// rank_ws.hip — the bf16 DNN3 rank kernel with the model weights stationary on the CU (see below).
#include "rank_mlp.hpp"

namespace pg {

// ---------------------------------------------------------------------------------------------
// dnn3_ws_kernel: DNN3 in bf16 with the weights STATIONARY on the CU.
// mlp_kernel streams all 384 KB of weights from L2 for every 128-item tile (768 KB per tile through the
// CU's L1 path: the waves wait on those loads for half their lifetime, profiles/history/r1g).  Here one
// persistent workgroup per CU — 4 waves, one per SIMD, 512 registers each — walks a contiguous range of
// 64-item tiles.  Wave w owns output columns 64w..64w+63 of layer 2 and keeps that 512 x 64 slice of W2 for the
// whole launch (62 MFMA B fragments in its AGPR half, 2 in LDS); it owns hidden columns 128w..128w+127 of
// layer 1: the fragments of the first 64 stay in LDS (64 KB for the workgroup), those of the other 64 (16 KB
// per wave and tile) are the only weights still streamed, requested a tile ahead in the head phase.  Two
// n-blocks per wave is the point of the shape: an A fragment read from LDS feeds two MFMAs, which keeps the LDS
// port at half load (one n-block per wave — 8 waves x 128 registers — was LDS-bound at 0.82 ms per 1.28 M
// items).  With one wave per SIMD nothing hides a memory latency but the code itself, so every load runs a
// phase ahead of its use: tile descriptors two tiles ahead, the candidate's row id one tile ahead, its table
// row during layer 2 of the previous tile; the request's layer-1 partial (c1) and the biases sit in LDS; a
// tile's scores are finished under the next tile's layer 1.  Layers 1 and 2 have mlp_kernel<1, 512, 256, …>'s
// arithmetic and k order; the head sums a lane's 32 columns, then the item's 8 partials in slot order (a fixed
// order of its own, inside the bf16 mode's 1e-5: DESIGN.md 5.2).
// ---------------------------------------------------------------------------------------------
constexpr size_t kWsRegion = (size_t)kWsItems * (kDIN + 512) * 2;         // X + H1 tiles (bf16)
constexpr size_t kWsW1L = 4 * 16 * 1024;     // resident half of W1: 16 fragments per wave
constexpr size_t kWsW2L = 4 * 2 * 1024;      // the two W2 fragments per wave that do not fit its AGPR half
constexpr size_t ws_lds_bytes() { return kWsRegion + kWsW1L + kWsW2L + (256 + 256 + 512 + 8 * kWsItems) * 4; }

// The MFMAs are issued as volatile asm: their order — and that of the LDS reads written between them — is then
// exactly the source order (the scheduler re-packed every builtin version of these pipelines into read → wait →
// MFMA), and the "a" constraint keeps W2 in the AGPR half with no copies.  The compiler does not know these are
// MFMAs: WS_MFMA_DONE supplies the wait states a VALU read of their results needs.
#define WS_MFMA_VV(acc, b, x) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(b), "v"(x))
#define WS_MFMA_AV(acc, b, x) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(b), "v"(x))
#define WS_MFMA_READY4(a0, a1, a2, a3) asm volatile("s_nop 3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3))
#define WS_MFMA_DONE4(a0, a1, a2, a3) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3))
__device__ __forceinline__ float ws_relu(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, __builtin_inff()); }

struct WsTile {
    uint32_t req, item0, cnt;
};

// MULTI: a model with several heads on the shared trunk (n_out <= kMaxHeads outputs, EasyrecResponse.multiValModule,
// algorithm/eas/easyrec_response.go:35-70).  Head 0 is the code above, bit for bit.  The other heads run the same
// relu → dot over the lane's 32 columns with their own w3 row (from L1 / L2: the LDS is full), and their 4 partials per
// item (one per wave) go through a per-workgroup block of GLOBAL scratch instead of LDS (the workgroup's own L1 keeps it coherent across
// the barrier); they are finished under the next tile's block A like head 0's — by the lanes that idle there: lane group
// g = lane / 16 finishes heads g and g + 4 of the wave's 16 items, so the sigmoids of up to four heads cost one.
template <bool MULTI>
__global__ __launch_bounds__(256, 1) void dnn3_ws_kernel(MlpArgs a) {
    constexpr int H1 = 512, H2 = 256, KS1 = kDIN / 16, KS2 = H1 / 16;      // 8 / 32 k-steps
    constexpr int XT_B = kWsItems * kDIN * 2;                             // 16 KiB
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const XT = smem;
    char* const H1T = smem + XT_B;
    char* const W1L = smem + kWsRegion;
    float* const w3s = reinterpret_cast<float*>(smem + kWsRegion + kWsW1L + kWsW2L);
    float* const b2s = w3s + H2;
    float* const c1s = b2s + H2;
    float* const hps = c1s + H1;                                          // head partials [8 slots][64 items]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    char* const w2l = smem + kWsRegion + kWsW1L + wave * 2048;
    const uint32_t n_tiles = *a.n_tiles;
    const uint32_t t_begin = (uint32_t)(((uint64_t)n_tiles * blockIdx.x) / gridDim.x);
    const uint32_t t_end = (uint32_t)(((uint64_t)n_tiles * (blockIdx.x + 1)) / gridDim.x);
    if (t_begin >= t_end) return;

    // layer-2 slice of this wave: B fragments of n-blocks 2*wave, 2*wave+1 — resident for the launch
    // (62 of the 64 fragments in the AGPR half; the two of the last k-step in LDS: the allocator needs a few
    // AGPRs of its own, and a fragment it spills comes back through scratch behind an s_waitcnt vmcnt(0))
    bf16x8 w2r[2][KS2 - 1];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
        for (int ks = 0; ks < KS2 - 1; ++ks)
            w2r[nb][ks] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(a.w2p) +
                                                           (size_t)((wave * 2 + nb) * KS2 + ks) * 1024 + lane * 16);
        *reinterpret_cast<uint4*>(w2l + nb * 1024 + lane * 16) = *reinterpret_cast<const uint4*>(
            reinterpret_cast<const char*>(a.w2p) + (size_t)((wave * 2 + nb) * KS2 + KS2 - 1) * 1024 + lane * 16);
    }
    // layer-1 fragments of this wave: 32 consecutive KiB; the first 16 (hidden columns 128w..128w+63) → LDS
    const char* const w1w = reinterpret_cast<const char*>(a.w1p) + (size_t)wave * 32 * 1024;
    char* const w1l = W1L + wave * 16 * 1024;
#pragma unroll
    for (int f = 0; f < 16; ++f)
        *reinterpret_cast<uint4*>(w1l + f * 1024 + lane * 16) = *reinterpret_cast<const uint4*>(w1w + f * 1024 + lane * 16);
    w3s[tid] = a.w3[tid];                                                // DNN3: one head row for all requests
    b2s[tid] = a.b2[tid];

    // gather role: 4 threads per item, 8 consecutive quads (128 B) each
    const int g_item = tid >> 2, g_q0 = (tid & 3) * 8;
    auto load_desc = [&](uint32_t t) {                     // uniform; held in SGPRs once it has arrived
        WsTile d{0, 0, 0};
        if (t < t_end) {
            d.req = a.tile_req[t];
            d.item0 = a.tile_item0[t];
            d.cnt = a.tile_cnt[t];
        }
        return d;
    };
    auto uniform = [](const WsTile& d) {
        return WsTile{(uint32_t)__builtin_amdgcn_readfirstlane(d.req), (uint32_t)__builtin_amdgcn_readfirstlane(d.item0),
                      (uint32_t)__builtin_amdgcn_readfirstlane(d.cnt)};
    };
    // the row id as stored (clamped where it is used, so that nothing waits for it here)
    auto load_rowid = [&](const WsTile& d, uint32_t item) -> uint32_t {
        if (d.cnt == 0) return 0;
        return a.cand_rows[d.item0 + (item < d.cnt ? item : d.cnt - 1)];
    };
    float4 xq[8];
    auto load_rows = [&](uint32_t row, uint32_t q0) {
        row = row < a.tab_rows ? row : a.tab_rows - 1;
        const float4* src = reinterpret_cast<const float4*>(a.tab + (size_t)row * kDIN) + q0;
#pragma unroll
        for (int j = 0; j < 8; ++j) xq[j] = src[j];
    };
    WsTile cur = uniform(load_desc(t_begin)), nxt = uniform(load_desc(t_begin + 1));
    load_rows(load_rowid(cur, g_item), g_q0);
    uint32_t c1_req = 0xffffffffu;
#ifdef PG_WS_PROFILE
    uint64_t ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = __builtin_readcyclecounter();
#define WS_MARK(i) { const uint64_t tn = __builtin_readcyclecounter(); ph[i] += tn - tp; tp = tn; }
#else
#define WS_MARK(i)
#endif

    // the streamed layer-1 fragments (hidden columns 128w+64..128w+127, used by block B): requested a tile ahead, in
    // the head phase — a global load costs its wave ~40 issue cycles in a VALU phase, 60-90 between MFMAs, and at
    // the top of the tile they were a pure stall in front of the barrier (measured with PG_WS_PROFILE)
    bf16x8 w1g[KS1][2];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
            w1g[ks][nb] = *reinterpret_cast<const bf16x8*>(w1w + (16 + nb * KS1 + ks) * 1024 + lane * 16);

    // The scores of a tile are finished — 8 head partials per item summed in slot order, sigmoid, store — under
    // block A of the NEXT tile's layer 1 (16 items per wave): the partials' LDS reads stand before its first MFMA,
    // the arithmetic floats between the MFMAs, the store follows its last one.  Done after the head it cost every
    // wave ~700 idle-pipe cycles per tile.  (hps is rewritten only after the next tile's layer 2: two barriers on.)
    WsTile fin{0, 0, 0};
    float fin_p[8];
    const uint32_t fin_item = (uint32_t)wave * (kWsItems / 4) + (lane & 15);
    auto finalize_read = [&]() {
#pragma unroll
        for (int s = 0; s < 8; ++s) fin_p[s] = hps[s * kWsItems + fin_item];
    };
    // (MULTI) this workgroup's partials of heads 1..: [head - 1][4 waves][64 items] — a wave's two column halves are
    // added in the wave (lanes i and i + 32) before they are stored, so a head's score is b3 + s0 + s1 + s2 + s3
    float* const gp = MULTI ? a.head_part + (size_t)blockIdx.x * ((kMaxHeads - 1) * 4 * kWsItems) : nullptr;
    const uint32_t n_out = MULTI ? a.n_out : 1u;
    // (MULTI) lane group g = lane / 16 finishes heads g and g + 4 of the wave's 16 items
    float fin_q[4], fin_b = 0.0f;
    auto finalize_read_multi = [&]() {
        if constexpr (MULTI) {
            // (per-lane values from an opaque thread id: carried across the tile loop they are spilled)
            uint32_t l_ = threadIdx.x;
            asm volatile("" : "+v"(l_));
            const uint32_t grp = (l_ >> 4) & 3, item = (l_ >> 6) * (kWsItems / 4) + (l_ & 15);
            if (grp >= 1 && grp < n_out && fin.cnt) {
                fin_b = a.b3v[grp];
                const float* const src = a.head_part + (size_t)blockIdx.x * ((kMaxHeads - 1) * 4 * kWsItems) +
                                         (grp - 1) * 4 * kWsItems + item;
#pragma unroll
                for (int s = 0; s < 4; ++s) fin_q[s] = src[s * kWsItems];
            }
        }
    };
    auto finalize_write = [&]() {
        float z = a.b3;
#pragma unroll
        for (int s = 0; s < 8; ++s) z += fin_p[s];
        if (lane < kWsItems / 4 && fin_item < fin.cnt) a.out[fin.item0 + fin_item] = 1.0f / (1.0f + expf(-z));
    };
    // (MULTI) the previous tile's other heads: read before the barrier behind layer 1 (every wave's reads are complete
    // there, this tile's head phase may then rewrite the block), finished behind layer 2, whose MFMAs cover the loads
    auto finalize_write_multi = [&]() {
        if constexpr (MULTI) {
            uint32_t l_ = threadIdx.x;
            asm volatile("" : "+v"(l_));
            const uint32_t grp = (l_ >> 4) & 3, item = (l_ >> 6) * (kWsItems / 4) + (l_ & 15);
            if (grp >= 1 && grp < n_out && item < fin.cnt) {
                float zo = fin_b;
#pragma unroll
                for (int s = 0; s < 4; ++s) zo += fin_q[s];
                a.out[(size_t)grp * a.out_stride + fin.item0 + item] = 1.0f / (1.0f + expf(-zo));
            }
        }
    };
    // heads 4..7 (rare): read and finished in one go, the latency shows
    auto finalize_high_heads = [&]() {
        if constexpr (MULTI) {
            uint32_t l_ = threadIdx.x;
            asm volatile("" : "+v"(l_));
            const uint32_t o = ((l_ >> 4) & 3) + 4, item = (l_ >> 6) * (kWsItems / 4) + (l_ & 15);
            if (o < n_out && item < fin.cnt) {
                float zo = a.b3v[o];
                const float* const src = a.head_part + (size_t)blockIdx.x * ((kMaxHeads - 1) * 4 * kWsItems) +
                                         (o - 1) * 4 * kWsItems + item;
#pragma unroll
                for (int s = 0; s < 4; ++s) zo += src[s * kWsItems];
                a.out[(size_t)o * a.out_stride + fin.item0 + item] = 1.0f / (1.0f + expf(-zo));
            }
        }
    };

    for (uint32_t tile = t_begin; tile < t_end; ++tile) {
        // ---- requests issued a phase (or more) ahead of their use
        uint32_t tid_o = tid;                              // opaque copy: what derives from it is recomputed per
        asm volatile("" : "+v"(tid_o));                    // tile instead of being carried (and spilled) across it
        const WsTile nn = load_desc(tile + 2);
        const uint32_t nxt_row = load_rowid(nxt, tid_o >> 2);
        const uint32_t lane_off = (tid_o & 63) * 16;
        // ---- X tile from the prefetched rows; the request's layer-1 partial when the request changes
#pragma unroll
        for (int j = 0; j < 8; ++j) store_x_quad<1>(XT, tid_o >> 2, (tid_o & 3) * 8 + j, xq[j]);
        if (cur.req != c1_req) {
            c1_req = cur.req;
            *reinterpret_cast<float2*>(c1s + 2 * tid_o) =
                *reinterpret_cast<const float2*>(a.c1 + (size_t)cur.req * a.c1_stride + 2 * tid_o);
        }
        WS_MARK(0)
        __syncthreads();
        WS_MARK(1)

        // ---- layer 1: hidden columns [128*wave, +128) as two blocks of 64 (A: fragments from LDS, B: the
        // streamed ones), both 32-item blocks each; transposed accumulators (a lane owns 4 consecutive columns
        // of one item).  Operands are read one k-step ahead of the MFMAs that use them — with one wave per SIMD
        // nothing else hides the LDS latency — and block A's H1 stores ride between block B's MFMAs.
        {
            // (addresses from the opaque thread id: loop-invariant ones get hoisted out of the tile loop and spilled)
            const int i32 = tid_o & 31, h = (tid_o >> 5) & 1, sw = tid_o & 15;
            const char* const x0 = XT + i32 * 256;
            const char* const x1 = XT + (32 + i32) * 256;
            f32x16 accA[2][2], accB[2][2];
            bf16x8 af[2][2], bl[2][2];
            auto init = [&](f32x16 (&acc)[2][2], int col0) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 cv = *reinterpret_cast<const float4*>(c1s + col0 + nb * 32 + 8 * g + 4 * h);
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) {
                            acc[mb][nb][4 * g + 0] = cv.x;
                            acc[mb][nb][4 * g + 1] = cv.y;
                            acc[mb][nb][4 * g + 2] = cv.z;
                            acc[mb][nb][4 * g + 3] = cv.w;
                        }
                    }
            };
            auto store_h = [&](const f32x16 (&acc)[2][2], int col0, int mb, int nb, int g) {
                store_h_quad<1, H1>(H1T, mb * 32 + i32, col0 + nb * 32 + 8 * g + 4 * h, ws_relu(acc[mb][nb][4 * g + 0]),
                                    ws_relu(acc[mb][nb][4 * g + 1]), ws_relu(acc[mb][nb][4 * g + 2]),
                                    ws_relu(acc[mb][nb][4 * g + 3]));
            };
            // block A
            init(accA, wave * 128);
            WS_MFMA_READY4(accA[0][0], accA[0][1], accA[1][0], accA[1][1]);
            af[0][0] = *reinterpret_cast<const bf16x8*>(x0 + ((h ^ sw) << 4));
            af[0][1] = *reinterpret_cast<const bf16x8*>(x1 + ((h ^ sw) << 4));
            bl[0][0] = *reinterpret_cast<const bf16x8*>(w1l + lane_off);
            bl[0][1] = *reinterpret_cast<const bf16x8*>(w1l + KS1 * 1024 + lane_off);
            finalize_read();                               // previous tile's head partials (fin.cnt = 0 on the first)
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                if (ks + 1 < KS1) {
                    af[(ks + 1) & 1][0] = *reinterpret_cast<const bf16x8*>(x0 + ((((ks + 1) * 2 + h) ^ sw) << 4));
                    af[(ks + 1) & 1][1] = *reinterpret_cast<const bf16x8*>(x1 + ((((ks + 1) * 2 + h) ^ sw) << 4));
                    bl[(ks + 1) & 1][0] = *reinterpret_cast<const bf16x8*>(w1l + (ks + 1) * 1024 + lane_off);
                    bl[(ks + 1) & 1][1] = *reinterpret_cast<const bf16x8*>(w1l + (KS1 + ks + 1) * 1024 + lane_off);
                } else {
                    af[0][0] = *reinterpret_cast<const bf16x8*>(x0 + ((h ^ sw) << 4));       // block B's first step
                    af[0][1] = *reinterpret_cast<const bf16x8*>(x1 + ((h ^ sw) << 4));
                }
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) WS_MFMA_VV(accA[mb][nb], bl[ks & 1][nb], af[ks & 1][mb]);
            }
            finalize_write();
            // block B, with block A's stores between its k-steps (A's last MFMA is >= 4 MFMAs old by the first)
            init(accB, wave * 128 + 64);
            WS_MFMA_READY4(accB[0][0], accB[0][1], accB[1][0], accB[1][1]);
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                if (ks + 1 < KS1) {
                    af[(ks + 1) & 1][0] = *reinterpret_cast<const bf16x8*>(x0 + ((((ks + 1) * 2 + h) ^ sw) << 4));
                    af[(ks + 1) & 1][1] = *reinterpret_cast<const bf16x8*>(x1 + ((((ks + 1) * 2 + h) ^ sw) << 4));
                }
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) WS_MFMA_VV(accB[mb][nb], w1g[ks][nb], af[ks & 1][mb]);
                if (ks == 0) WS_MFMA_DONE4(accA[0][0], accA[0][1], accA[1][0], accA[1][1]);
#pragma unroll
                for (int q = 2 * ks; q < 2 * ks + 2; ++q) store_h(accA, wave * 128, q >> 3, (q >> 2) & 1, q & 3);
            }
            // (tried: block B one 32-item block after the other, its first half's stores under the second half's
            // MFMAs — 440 vs 433 K cycles for layer 1: two accumulators alternating cost what the hidden stores won)
            WS_MFMA_DONE4(accB[0][0], accB[0][1], accB[1][0], accB[1][1]);
#pragma unroll
            for (int q = 0; q < 16; ++q) store_h(accB, wave * 128 + 64, q >> 3, (q >> 2) & 1, q & 3);
        }
        WS_MARK(2)
        finalize_read_multi();
        if (MULTI && n_out > 4) finalize_high_heads();
        __syncthreads();
        WS_MARK(3)

        // ---- layer 2: output columns [64*wave, +64) for both item blocks, W2 from the AGPR half, A fragments
        // two k-steps ahead.  The next tile's candidate rows are requested first.
        f32x16 acc2[2][2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4*>(b2s + wave * 64 + nb * 32 + 4 * h + 8 * g);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    acc2[mb][nb][4 * g + 0] = bv.x;
                    acc2[mb][nb][4 * g + 1] = bv.y;
                    acc2[mb][nb][4 * g + 2] = bv.z;
                    acc2[mb][nb][4 * g + 3] = bv.w;
                }
            }
        {
            const int i32 = tid_o & 31, h = (tid_o >> 5) & 1, sw = tid_o & 15;
            const char* const h1r0 = H1T + i32 * (H1 * 2);
            const char* const h1r1 = H1T + (32 + i32) * (H1 * 2);
            // (a valid row when there is no next tile: the loads below are unconditional)
            const uint32_t grow = nxt.cnt ? (nxt_row < a.tab_rows ? nxt_row : a.tab_rows - 1) : 0;
            const float4* const gsrc = reinterpret_cast<const float4*>(a.tab + (size_t)grow * kDIN) + (tid_o & 3) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) xq[j] = gsrc[j];
            bf16x8 af[3][2], wl[2];
            WS_MFMA_READY4(acc2[0][0], acc2[0][1], acc2[1][0], acc2[1][1]);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                af[s][0] = *reinterpret_cast<const bf16x8*>(h1r0 + (((s * 2 + h) ^ sw) << 4));
                af[s][1] = *reinterpret_cast<const bf16x8*>(h1r1 + (((s * 2 + h) ^ sw) << 4));
            }
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                if (ks + 2 < KS2) {
                    af[(ks + 2) % 3][0] = *reinterpret_cast<const bf16x8*>(h1r0 + ((((ks + 2) * 2 + h) ^ sw) << 4));
                    af[(ks + 2) % 3][1] = *reinterpret_cast<const bf16x8*>(h1r1 + ((((ks + 2) * 2 + h) ^ sw) << 4));
                }
                if (ks + 2 == KS2 - 1) {
                    wl[0] = *reinterpret_cast<const bf16x8*>(w2l + lane_off);
                    wl[1] = *reinterpret_cast<const bf16x8*>(w2l + 1024 + lane_off);
                }
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        if (ks < KS2 - 1) WS_MFMA_AV(acc2[mb][nb], w2r[nb][ks], af[ks % 3][mb]);
                        else WS_MFMA_VV(acc2[mb][nb], wl[nb], af[ks % 3][mb]);
                    }
            }
            WS_MFMA_DONE4(acc2[0][0], acc2[0][1], acc2[1][0], acc2[1][1]);
        }
        finalize_write_multi();
        WS_MARK(4)

        // next tile's streamed fragments (this tile's were last read by block B)
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                w1g[ks][nb] = *reinterpret_cast<const bf16x8*>(w1w + (16 + nb * KS1 + ks) * 1024 + lane_off);

        // ---- relu → dot head, from the accumulators: a lane owns 32 of its item's 256 h2 columns — its
        // partial runs over them in ascending order, the 8 partials of an item (wave, h) are then added in
        // slot order: z = (((b3 + p0) + p1) + …) + p7.  (mlp_kernel's order is two 128-column chains; bf16
        // scores are specified to 1e-5 either way — the MFMA's own accumulation order is not defined.)
        {
            float4 wv[2][4];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    wv[nb][g] = *reinterpret_cast<const float4*>(w3s + wave * 64 + nb * 32 + 8 * g + 4 * h);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                float p = 0.0f;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float v0 = acc2[mb][nb][4 * g + 0], v1 = acc2[mb][nb][4 * g + 1];
                        const float v2 = acc2[mb][nb][4 * g + 2], v3 = acc2[mb][nb][4 * g + 3];
                        p = __fmaf_rn(ws_relu(v0), wv[nb][g].x, p);
                        p = __fmaf_rn(ws_relu(v1), wv[nb][g].y, p);
                        p = __fmaf_rn(ws_relu(v2), wv[nb][g].z, p);
                        p = __fmaf_rn(ws_relu(v3), wv[nb][g].w, p);
                    }
                hps[(wave * 2 + ((tid_o >> 5) & 1)) * kWsItems + mb * 32 + (tid_o & 31)] = p;
            }
            if constexpr (MULTI) {
                // the other heads: same chain over the same columns, w3 row o from global memory (every tile reads the
                // same 1 KB per head: L1-resident), partials to the workgroup's global block
                for (uint32_t o = 1; o < n_out; ++o) {
                    const float* const w3o = a.w3 + (size_t)o * H2 + wave * 64 + 4 * h;
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) wv[nb][g] = *reinterpret_cast<const float4*>(w3o + nb * 32 + 8 * g);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        float p = 0.0f;
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                p = __fmaf_rn(ws_relu(acc2[mb][nb][4 * g + 0]), wv[nb][g].x, p);
                                p = __fmaf_rn(ws_relu(acc2[mb][nb][4 * g + 1]), wv[nb][g].y, p);
                                p = __fmaf_rn(ws_relu(acc2[mb][nb][4 * g + 2]), wv[nb][g].z, p);
                                p = __fmaf_rn(ws_relu(acc2[mb][nb][4 * g + 3]), wv[nb][g].w, p);
                            }
                        p += __shfl_xor(p, 32);
                        if (((tid_o >> 5) & 1) == 0) gp[((o - 1) * 4 + wave) * kWsItems + mb * 32 + (tid_o & 31)] = p;
                    }
                }
            }
        }
        WS_MARK(5)
        __syncthreads();                                   // partials visible; also: everyone is done with H1
        WS_MARK(6)
        WS_MARK(7)
        fin = cur;                                         // its scores are finished under the next tile's layer 1
        cur = nxt;
        nxt = uniform(nn);
    }
    finalize_read();
    finalize_write();
    finalize_read_multi();
    finalize_write_multi();
    if (MULTI && n_out > 4) finalize_high_heads();
#ifdef PG_WS_PROFILE
    if (lane == 0 && blockIdx.x < 4) {
        uint64_t* o = (uint64_t*)(a.field_emb) + (blockIdx.x * 4 + wave) * 8;
        for (int i = 0; i < 8; ++i) o[i] = ph[i];
    }
#endif
}

int launch_dnn3_ws(pg_ctx* ctx, const MlpArgs& a) {
    constexpr size_t lds = ws_lds_bytes();
    int rc;
    if (a.n_out > 1) {
        if ((rc = ensure_dyn_lds(ctx, (const void*)dnn3_ws_kernel<true>, lds))) return rc;
        dnn3_ws_kernel<true><<<ctx->num_cus, 256, lds, ctx->stream>>>(a);
        return PG_OK;
    }
    if ((rc = ensure_dyn_lds(ctx, (const void*)dnn3_ws_kernel<false>, lds))) return rc;
#ifdef PG_WS_PROFILE
    static uint64_t* dbg = nullptr;
    if (!dbg) hipMalloc(&dbg, 4 * 4 * 8 * 8);
    MlpArgs b = a;
    b.field_emb = reinterpret_cast<const float* const*>(dbg);
    dnn3_ws_kernel<false><<<ctx->num_cus, 256, lds, ctx->stream>>>(b);
    uint64_t hcyc[128];
    hipMemcpy(hcyc, dbg, sizeof hcyc, hipMemcpyDeviceToHost);
    static int calls = 0;
    if (++calls == 3)
        for (int wv = 0; wv < 8; ++wv) {
            fprintf(stderr, "ws wg %d wave %d:", wv / 4, wv % 4);
            for (int i = 0; i < 8; ++i) fprintf(stderr, " %8llu", (unsigned long long)hcyc[wv * 8 + i]);
            fprintf(stderr, "\n");
        }
#else
    dnn3_ws_kernel<false><<<ctx->num_cus, 256, lds, ctx->stream>>>(a);
#endif
    return PG_OK;
}

}  // namespace pg
