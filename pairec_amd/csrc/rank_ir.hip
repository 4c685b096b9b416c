// rank_ir.hip — FM + two-tower rank over materialised item records (BASELINE.json configs[3]) with the towers'
// weights stationary in registers, producer / consumer waves, and the records arriving three tiles ahead by LDS-DMA.
#include "rank_mlp.hpp"

namespace pg {

// ---------------------------------------------------------------------------------------------
// fm2t_irs_kernel: the item side of the FM + two-tower model (algorithm/eas/fm_request.go:29-79 builds the request the
// reference sends to PAI-EAS; service/rank/rank_service.go:264-289 calls it per batch) for the benchmark's shape —
// 8 item fields x 16, item tower 128 -> 256 -> 64, bf16 — over pg_fm2t_item_rows_build's 640-B records.
//
// mlp_kernel<..., MODEL 3> (rank_mlp.hip) spends a tile as: tile descriptor -> candidate row -> record (three dependent
// round trips), the towers' 96 KB of weight fragments re-read from L2 for every 128-item tile, five barriers; two
// workgroups per CU hide part of it: 0.315 ms per 1.28 M items = 0.27 of the HBM bound on 544 B per item, its own
// all-hits floor 0.20 ms.  Here ONE persistent workgroup of EIGHT waves per CU (two per SIMD) walks a
// contiguous range of 64-item tiles:
//   * waves 0-3 are CONSUMERS — all the MFMAs: wave w keeps the sixteen layer-1 fragments of hidden columns 64 w .. + 63
//     (both item blocks: an A fragment read from LDS feeds two MFMAs, four accumulator chains) and the sixteen layer-2
//     fragments of its output block, 96 KB of weights in registers, no per-tile weight traffic at all —, waves 4-7
//     PRODUCERS — no MFMA: while the consumers run layer 1 of tile t, waves 6 / 7 finish tile t - 1 (head); while they run
//     layer 2, every producer turns its share of tile t + 1's records into the (double-buffered) X tile and the FM terms and
//     issues the DMAs of tile t + 3.  A SIMD holds one wave of each kind, so its matrix pipe and its vector ALU work on
//     different tiles at the same time;
//   * every vector-memory load of the loop is an LDS-DMA (`global_load_lds`, inline asm: the compiler must not know — it
//     would guard every LDS access with a vmcnt(0)), issued THREE tiles before its data is used and awaited with counted
//     `s_waitcnt vmcnt(N)`, never 0: a record is fetched by FOUR adjacent lanes of a producer (lane j: quad j of every field,
//     eight 16-B pieces; lanes 0 / 1 the two quads of linear weights) into the thread's own lane-linear spot of one of two
//     raw slots and read back from there by the same thread — the FM chains then run inside one lane, in the specification's
//     order, with no idle lanes; the request's FM prefix and user-tower output go straight to their LDS slots; tile
//     descriptors and candidate rows come through the scalar cache (constant address space), a tile ahead of the DMA that
//     needs them;
//   * barriers are bare `s_barrier`s behind an LDS-only wait; two per tile;
//   * the waits count events per wave: every producer has ten per tile — waves 4 / 5 nine record DMAs and one request DMA
//     (FM prefix / tower output), waves 6 / 7 nine record DMAs and the head's store, which every lane issues on every tile
//     (invalid items store to a sink) so that the count never varies.
// Arithmetic: exactly mlp_kernel<1, 256, 64, false, ..., MODEL 3>'s — the FM sums in the specification's order (user
// prefix first, fields ascending), t_k = fmaf(s_k, s_k, -q_k), the balanced tree over k (levels 1-2 in the lane, 3-4
// across the record's four lanes), lin + 0.5 * cross with the linear weights added one by one (lane 0 the first four,
// lane 1 the rest); both layers' MFMA sequences k-ascending from the same bias-initialised accumulators; the head's two
// 32-column half chains.  Scores are bit-identical to that kernel's, hence to the per-field path's
// (test_fm2t_materialised_item_records_are_bit_identical).
// ---------------------------------------------------------------------------------------------
constexpr int kIrTH = 256, kIrTO = 64;       // item tower widths
constexpr int kIrHS = kIrTO + 4;             // fp32 H2 tile row stride (floats): an odd number of 16-B quads per row
constexpr int kIrSlots = 6;                  // per-tile request data in LDS: ring of six tiles (written three ahead, read one behind)
constexpr int kIrSlotF = kIrsItems + kIrTO + 48;                // floats per ring slot: FM terms | tower output | FM prefix
constexpr size_t kIrWaveRaw = 9 * 1024;                         // a producer wave's share of a raw slot: 8 field pieces + the linear quads
constexpr size_t kIrRaw = 4 * kIrWaveRaw;                       // 36 KiB per raw slot
constexpr size_t kIrXT = (size_t)kIrsItems * kDIN * 2;          // 16 KiB
constexpr size_t kIrH1 = (size_t)kIrsItems * kIrTH * 2;         // 32 KiB
constexpr size_t kIrH2 = (size_t)kIrsItems * kIrHS * 4;         // 17 KiB per H2 tile
constexpr size_t kIrMeta = 4 * 2 * 64 * 4;                        // per producer wave two slots of 64 dwords: rows + next descriptor
constexpr size_t ir_lds_bytes() { return 2 * kIrRaw + 2 * kIrXT + kIrH1 + kIrH2 + (size_t)kIrSlots * kIrSlotF * 4 + kIrTO * 4 + kIrMeta; }
static_assert(ir_lds_bytes() <= 160 * 1024, "fm2t_irs_kernel: LDS budget");

struct IrTile {
    uint32_t req, item0, cnt;
};

// every lane the value of the lane one below (row_shr:1; lane 0 of a row of 16 keeps its own)
__device__ __forceinline__ float ir_from_lane_minus1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, false));
}
__device__ __forceinline__ float ir_lane_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float ir_lane_xor2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
}
// workgroup barrier that waits for this wave's LDS / scalar traffic only (the DMAs stay in flight)
__device__ __forceinline__ void ir_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N>
__device__ __forceinline__ void ir_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
// One LDS-DMA instruction: every active lane's 16 (4) bytes at its own global address -> LDS byte address `lds_addr` (wave-
// uniform) + lane * 16 (4).  Inline asm, as csrc/recall.hip's dma_one: hipcc does not know these are loads, so the counted
// waits are the only ones.  M0 carries the LDS address; it is a reserved register that nothing else in this kernel uses.
__device__ __forceinline__ void ir_dma16(const void* g, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void ir_dma4(const void* g, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(g), "s"(lds_addr) : "memory");
}
// max(v, +0) in ONE instruction (fmaxf compiles to a canonicalising v_max v, v in front of it); NaN -> 0, -0 -> +0, as the
// ternary of mlp_kernel
__device__ __forceinline__ float ir_relu(float v) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
    return r;
}

__global__ __launch_bounds__(512, 1) void fm2t_irs_kernel(MlpArgs a) {
    constexpr int KS1 = kDIN / 16, KS2 = kIrTH / 16;                       // 8 / 16 k-steps
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const RAW0 = smem;                                                // raw record slots [2]
    char* const XT0 = smem + 2 * kIrRaw;                                    // X tiles [2]
    char* const H1T = XT0 + 2 * kIrXT;
    float* const H2T = reinterpret_cast<float*>(H1T + kIrH1);               // H2 tile
    float* const ring = H2T + kIrsItems * kIrHS;                            // [slot]: b3s[64] | w3s[64] | fus[48]
    float* const b2s = ring + kIrSlots * kIrSlotF;                          // ib2[64]
    const uint32_t* const meta = reinterpret_cast<const uint32_t*>(b2s + kIrTO);   // [producer][2][64]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);   // LDS byte address of smem
    const uint32_t lds_ring = lds0 + (uint32_t)(2 * kIrRaw + 2 * kIrXT + kIrH1 + kIrH2);
    const uint32_t lds_meta = lds_ring + (uint32_t)(kIrSlots * kIrSlotF * 4 + kIrTO * 4);
    const uint32_t n_tiles = *a.n_tiles;
    const uint32_t t_begin = (uint32_t)(((uint64_t)n_tiles * blockIdx.x) / gridDim.x);
    const uint32_t t_end = (uint32_t)(((uint64_t)n_tiles * (blockIdx.x + 1)) / gridDim.x);
    if (t_begin >= t_end) return;

    // ---- the towers, for the whole launch
    const int nbp = wave & 3;                                               // consumer w: hidden n-blocks 2w, 2w + 1 of layer 1 ...
    bf16x8 w1r[2][KS1];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
            w1r[nb][ks] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(a.w1p) + (size_t)((nbp * 2 + nb) * KS1 + ks) * 1024 + lane * 16);
    const int mb2 = wave & 1, nb2 = (wave >> 1) & 1;                        // ... and output block (mb2, nb2) of layer 2
    bf16x8 w2r[KS2];
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks)
        w2r[ks] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(a.w2p) + (size_t)(nb2 * KS2 + ks) * 1024 + lane * 16);
    f32x16 c1v[2];                                                          // layer 1's accumulator start values: the ib1 columns of this lane
    {
        const int h_ = lane >> 5;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const float4 cv = *reinterpret_cast<const float4*>(a.c1 + (nbp * 2 + nb) * 32 + 4 * h_ + 8 * g);
                c1v[nb][4 * g + 0] = cv.x; c1v[nb][4 * g + 1] = cv.y; c1v[nb][4 * g + 2] = cv.z; c1v[nb][4 * g + 3] = cv.w;
            }
        }
    }
    asm volatile("" : "+v"(c1v[0]), "+v"(c1v[1]));
    if (tid < kIrTO) b2s[tid] = a.b2[tid];
    // (the loads above are ordinary ones: their wait belongs here, not at their first use inside the loop, where the
    // compiler would repeat a vmcnt(0) on every trip)
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) asm volatile("" : "+v"(w1r[0][ks]), "+v"(w1r[1][ks]));
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) asm volatile("" : "+v"(w2r[ks]));

    // ---- tile descriptors and candidate rows (producers).  In the loop they arrive like everything else — by LDS-DMA, a
    // tile before they are needed, counted in the same queue: ONE `global_load_lds_dword` per wave and tile brings the wave's
    // candidate rows of tile T (every lane its own record's row; the four lanes of a record fetch the same dword) and, in
    // lanes 1..3, tile T + 1's descriptor, into the wave's meta slot.  (Scalar loads here cost the producers two dependent
    // round trips per tile with nothing to overlap them: 1 200 of a tile's 6 400 cycles.)  Only the prologue reads the
    // tables through the scalar cache.
    typedef const __attribute__((address_space(4))) uint32_t* cu32p;
    const cu32p k_req = (cu32p)(uintptr_t)a.tile_req, k_item0 = (cu32p)(uintptr_t)a.tile_item0, k_cnt = (cu32p)(uintptr_t)a.tile_cnt;
    const int pw = wave & 3;                                // producer index of waves 4..7
    auto desc = [&](uint32_t t) {
        const uint32_t tc = t < t_end ? t : t_end - 1;      // (past the range: a valid entry, cnt forced to 0)
        IrTile d{k_req[tc], k_item0[tc], k_cnt[tc]};
        if (t >= t_end) d.cnt = 0;
        return d;
    };
    // index into cand_rows of this lane's record of tile d (a request's last tile: clamped to its last candidate)
    auto cand_index = [&](const IrTile& d, uint32_t l_) {
        const uint32_t it = (uint32_t)pw * 16 + (l_ >> 2), last = d.cnt ? d.cnt - 1 : 0u;
        return d.item0 + (it < last ? it : last);
    };
    // the meta DMA: rows of tile d (lanes 0, 4.., and their duplicates), descriptor of tile `tn` (lanes 1..3) -> meta slot
    auto meta_issue = [&](uint32_t tn, const IrTile& d, uint32_t slot) {
        uint32_t l_ = (uint32_t)lane;
        asm volatile("" : "+v"(l_));
        const uint32_t tc = tn < t_end ? tn : t_end - 1;
        const uint32_t* p = a.cand_rows + cand_index(d, l_);
        p = l_ == 1 ? a.tile_req + tc : p;
        p = l_ == 2 ? a.tile_item0 + tc : p;
        p = l_ == 3 ? a.tile_cnt + tc : p;
        ir_dma4(p, lds_meta + (uint32_t)((pw * 2 + slot) * 64 * 4));
    };
    // ---- the DMAs of one tile into raw slot `sl` (records) and ring slot `rs` (wave 4: FM prefix, wave 5: tower output)
    auto issue = [&](uint32_t req, uint32_t rowA, uint32_t rowB, uint32_t rowC, uint32_t sl, uint32_t rs) {
        uint32_t l_ = (uint32_t)lane;
        asm volatile("" : "+v"(l_));                        // (per-lane values re-derived per tile: carried across the loop they are spilled)
        if (wave == 4) {
            // fus[L]: L < 16 the prefix's s, < 32 its q, 32 its linear part (lanes up to 47 land in the slot's padding)
            const uint32_t fi = l_ < 16 ? 1 + l_ : (l_ < 32 ? 1 + kFmMaxK + (l_ - 16) : 0u);
            if (l_ < 48) ir_dma4(a.fm_user + (size_t)req * kFmUserStride + fi, lds_ring + (rs * kIrSlotF + kIrsItems + kIrTO) * 4);
        } else if (wave == 5) {
            ir_dma4(a.w3 + (size_t)req * a.w3_stride + l_, lds_ring + (rs * kIrSlotF + kIrsItems) * 4);
        }
        // Records go out as whole 128-B lines: EIGHT adjacent lanes per record and instruction (an instruction touches 8 lines
        // instead of 16 half lines — the address path's cost is per line: 0.260 -> 0.219 ms), two passes of eight records x
        // the four lines of embeddings; the fifth line's two quads of linear weights with four lanes per record, sixteen
        // records at once.  The wave's raw share is then: piece 4 p + i = line i of records 8 p .. 8 p + 7 (record g at
        // g * 128), piece 8 the linear quads (record r at r * 64)
        auto clampr = [&](uint32_t r) { return r < a.irow_count ? r : a.irow_count; };   // (outside the store: the defaults' record)
        // (the two 64-B halves of a line change places for records 2, 3, 6, 7 of a pass — on the SOURCE side, the LDS image of
        // a DMA is lane-linear —: the conversion's 16-lane read groups hold records {0, 3, 5, 6} / {1, 2, 4, 7}, whose
        // quads would otherwise share banks two by two)
        const uint32_t jj = (l_ & 7) ^ ((l_ >> 2) & 4);
        const char* const recA = reinterpret_cast<const char*>(a.irows) + (size_t)clampr(rowA) * (kItemRowFloats * 4) + jj * 16;
        const char* const recB = reinterpret_cast<const char*>(a.irows) + (size_t)clampr(rowB) * (kItemRowFloats * 4) + jj * 16;
        const uint32_t dst = lds0 + sl * (uint32_t)kIrRaw + (uint32_t)pw * (uint32_t)kIrWaveRaw;
#pragma unroll
        for (int i = 0; i < 4; ++i) ir_dma16(recA + i * 128, dst + i * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) ir_dma16(recB + i * 128, dst + (4 + i) * 1024);
        const uint32_t j = l_ & 3;
        if (j < 2) ir_dma16(reinterpret_cast<const char*>(a.irows) + (size_t)clampr(rowC) * (kItemRowFloats * 4) + kDIN * 4 + j * 16, dst + 8 * 1024);
    };
    // ---- raw slot `sl` (this thread's own pieces) -> X tile and FM terms of the tile; the ring slot holds its request's FM prefix
    auto convert = [&](uint32_t sl, uint32_t rs, char* XT) {
        uint32_t t_ = (uint32_t)tid;
        asm volatile("" : "+v"(t_));
        const int j = t_ & 3;
        const int r = (int)((t_ - 256) >> 2);
        const char* const wbase = RAW0 + sl * kIrRaw + (size_t)pw * kIrWaveRaw;
        const uint32_t rw = (t_ & 63) >> 2;                  // record within the wave: its lines were fetched in pass rw / 8, group rw % 8
        const char* const wraw = wbase + (rw >> 3) * 4096 + (rw & 7) * 128 + j * 16;
        const char* const weven = wraw + ((rw >> 1) & 1) * 64;          // even fields' half of the record's lines (see issue)
        const char* const wodd = wraw + (((rw >> 1) & 1) ^ 1) * 64;
        float4 e[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) e[f] = *reinterpret_cast<const float4*>(((f & 1) ? wodd : weven) + (f >> 1) * 1024);
        const float4 lq = *reinterpret_cast<const float4*>(wbase + 8 * 1024 + (t_ & 63) * 16);   // (lanes 0 / 1 of the record)
        const float* fu = ring + rs * kIrSlotF + kIrsItems + kIrTO;
        // the eight chains of this lane (s and q of four columns) as plain v_add_f32 / v_fma_f32: beside another wave's MFMAs
        // on the SIMD a packed fp32 instruction costs more than the two it replaces (asm: the compiler would pack them again)
        float sc[4], qc[4];
        {
            const float4 s4 = *reinterpret_cast<const float4*>(fu + 4 * j);
            const float4 q4 = *reinterpret_cast<const float4*>(fu + 16 + 4 * j);
            sc[0] = s4.x; sc[1] = s4.y; sc[2] = s4.z; sc[3] = s4.w;
            qc[0] = q4.x; qc[1] = q4.y; qc[2] = q4.z; qc[3] = q4.w;
        }
#pragma unroll
        for (int f = 0; f < 8; ++f) {                       // user prefix first, fields ascending
            const float xv[4] = {e[f].x, e[f].y, e[f].z, e[f].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                asm("v_add_f32 %0, %1, %0" : "+v"(sc[c]) : "v"(xv[c]));
                asm("v_fma_f32 %0, %1, %1, %0" : "+v"(qc[c]) : "v"(xv[c]));
            }
        }
        float s_[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) s_[c] = __fmaf_rn(sc[c], sc[c], -qc[c]);
        float cross = (s_[0] + s_[1]) + (s_[2] + s_[3]);    // tree levels 1, 2 (columns of one quad)
        cross = cross + ir_lane_xor1(cross);                // level 3: quads 2m, 2m + 1
        cross = cross + ir_lane_xor2(cross);                // level 4
        // linear term: prefix + the eight weights one by one — lane 0 adds 0..3, lane 1 (which starts from lane 0's sum) 4..7
        float lin = fu[32];
        lin = lin + lq.x; lin = lin + lq.y; lin = lin + lq.z; lin = lin + lq.w;
        float lin1 = ir_from_lane_minus1(lin);
        lin1 = lin1 + lq.x; lin1 = lin1 + lq.y; lin1 = lin1 + lq.z; lin1 = lin1 + lq.w;
        if (j == 1) ring[rs * kIrSlotF + r] = lin1 + 0.5f * cross;
#pragma unroll
        for (int f = 0; f < 8; ++f) store_x_quad<1>(XT, r, f * 4 + j, e[f]);
    };
    // ---- the head of a tile whose H2 tile / FM terms / tower output are in LDS (waves 6, 7: two threads per item).  The
    // store is issued by every lane for every tile: items past the tile's count (and the odd lanes) write to the sink
    auto head = [&](const IrTile& d, uint32_t rs) {
        uint32_t t_ = (uint32_t)tid;
        asm volatile("" : "+v"(t_));
        const uint32_t u = t_ - 384;
        const int row = (int)(u >> 1), half = (int)(u & 1);
        const float4* hr = reinterpret_cast<const float4*>(H2T + row * kIrHS + half * (kIrTO / 2));
        const float4* wr = reinterpret_cast<const float4*>(ring + rs * kIrSlotF + kIrsItems + half * (kIrTO / 2));
        float4 x[kIrTO / 8], y[kIrTO / 8];                  // (all sixteen reads in flight before the chain starts)
#pragma unroll
        for (int m = 0; m < kIrTO / 8; ++m) {
            x[m] = hr[m];
            y[m] = wr[m];
        }
        float p = half ? 0.0f : ring[rs * kIrSlotF + row];
#pragma unroll
        for (int m = 0; m < kIrTO / 8; ++m) {
            p = __fmaf_rn(x[m].x, y[m].x, p);
            p = __fmaf_rn(x[m].y, y[m].y, p);
            p = __fmaf_rn(x[m].z, y[m].z, p);
            p = __fmaf_rn(x[m].w, y[m].w, p);
        }
        const float o = ir_lane_xor1(p);
        const float z = half ? (o + p) : (p + o);
        float* const dst = (half == 0 && (uint32_t)row < d.cnt) ? a.out + d.item0 + row : a.sink + u;
        *dst = 1.0f / (1.0f + expf(-z));
    };
    auto ring_add = [](uint32_t s, uint32_t k) { const uint32_t x = s + k; return x >= (uint32_t)kIrSlots ? x - kIrSlots : x; };

    // ---- prologue (producers).  Tile t_begin synchronously through raw slot t_begin & 1; then tiles + 1 and + 2 in flight
    // with the event sequence the loop's counted waits assume: [DMAs of + 1] [waves 6 / 7: one store] [meta] [DMAs of + 2].
    // The descriptors of tiles t - 1 .. t + 3 travel through the loop as wave-uniform values (dA .. d3), shifted per trip
    uint32_t s6 = t_begin % kIrSlots;                       // ring slot of tile t
    IrTile dA{0, 0, 0}, dB{0, 0, 0}, dC{0, 0, 0}, dD{0, 0, 0}, d3{0, 0, 0};
    uint32_t r0[3] = {0, 0, 0}, r1[3] = {0, 0, 0}, r2[3] = {0, 0, 0};
    if (wave >= 4) {
        dB = desc(t_begin);
        dC = desc(t_begin + 1);
        dD = desc(t_begin + 2);
        d3 = desc(t_begin + 3);
        dA = dB;
        dA.cnt = 0;                                         // (first trip's head: nothing to finish, all lanes to the sink)
        uint32_t l_ = (uint32_t)lane;
        asm volatile("" : "+v"(l_));
        // (lane l's three records of a tile: l / 8 and 8 + l / 8 for the line passes, l / 4 for the linear quads)
        const uint32_t la = (l_ >> 3) * 4, lb = 32 + (l_ >> 3) * 4;
        r0[0] = a.cand_rows[cand_index(dB, la)]; r0[1] = a.cand_rows[cand_index(dB, lb)]; r0[2] = a.cand_rows[cand_index(dB, l_)];
        r1[0] = a.cand_rows[cand_index(dC, la)]; r1[1] = a.cand_rows[cand_index(dC, lb)]; r1[2] = a.cand_rows[cand_index(dC, l_)];
        r2[0] = a.cand_rows[cand_index(dD, la)]; r2[1] = a.cand_rows[cand_index(dD, lb)]; r2[2] = a.cand_rows[cand_index(dD, l_)];
        issue(dB.req, r0[0], r0[1], r0[2], t_begin & 1, s6);
        ir_wait_vm<0>();
    }
    ir_barrier();                                           // the FM prefix (wave 4's DMA)
    if (wave >= 4) {
        convert(t_begin & 1, s6, XT0 + (t_begin & 1) * kIrXT);
        issue(dC.req, r1[0], r1[1], r1[2], (t_begin + 1) & 1, ring_add(s6, 1));
        if (wave >= 6) a.sink[128 + (tid - 384)] = 0.0f;
        meta_issue(t_begin + 4, d3, (t_begin + 1) & 1);     // rows of tile + 3, descriptor of tile + 4: read in the first trip
        issue(dD.req, r2[0], r2[1], r2[2], t_begin & 1, ring_add(s6, 2));
        if (wave == 4) ir_wait_vm<20>();                    // tile t_begin + 1's FM prefix, before the first barrier A
    }

#ifdef PG_IR_PROFILE
    uint64_t ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tpc = __builtin_readcyclecounter();
#define IR_MARK(i) { const uint64_t tn_ = __builtin_readcyclecounter(); ph[i] += tn_ - tpc; tpc = tn_; }
#else
#define IR_MARK(i)
#endif
    for (uint32_t t = t_begin; t < t_end; ++t) {
        uint32_t t_ = (uint32_t)tid;
        asm volatile("" : "+v"(t_));
        const int i32 = t_ & 31, h = (t_ >> 5) & 1, sw = t_ & 15;
        IR_MARK(7)
        ir_barrier();                                       // A: X (tile t) and H2 (tile t - 1) complete
        IR_MARK(0)
        if (wave < 4) {
            // ---- consumers: layer 1 of tile t — hidden columns 64 w .. + 63 for both item blocks
            const char* const XT = XT0 + (t & 1) * kIrXT;
            // item block 0's MFMAs, then item block 1's with block 0's relu -> bf16 -> H1 stores between them (the matrix pipe
            // runs block 1 while the vector ALU packs block 0); the accumulators start from ib1 through the first MFMA's C
            // operand (no copies)
            f32x16 acc[2][2];
            auto read_a = [&](int mb, int ks) {
                return *reinterpret_cast<const bf16x8*>(XT + (mb * 32 + i32) * 256 + (((ks * 2 + h) ^ sw) << 4));
            };
            auto store_block = [&](int mb, int nb, int g) {
                store_h_quad<1, kIrTH>(H1T, mb * 32 + i32, (nbp * 2 + nb) * 32 + 8 * g + 4 * h, ir_relu(acc[mb][nb][4 * g + 0]),
                                       ir_relu(acc[mb][nb][4 * g + 1]), ir_relu(acc[mb][nb][4 * g + 2]), ir_relu(acc[mb][nb][4 * g + 3]));
            };
            // the A fragments are read TWO steps ahead of the MFMAs that use them (with one consumer per SIMD nothing else
            // hides the LDS latency); steps 0..7 = item block 0, 8..15 = item block 1
            bf16x8 af[3];
            af[0] = read_a(0, 0);
            af[1] = read_a(0, 1);
#pragma unroll
            for (int st = 0; st < 2 * KS1; ++st) {
                const int mb = st >> 3, ks = st & 7;
                if (st + 2 < 2 * KS1) af[(st + 2) % 3] = read_a((st + 2) >> 3, (st + 2) & 7);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1r[nb][ks], af[st % 3], ks == 0 ? c1v[nb] : acc[mb][nb], 0, 0, 0);
                if (mb == 1) {                              // one quad of block 0 per step of block 1: packed while the pipe runs
                    __builtin_amdgcn_sched_barrier(0);
                    store_block(0, ks >> 2, ks & 3);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) store_block(1, q >> 2, q & 3);
            IR_MARK(1)
        } else {
            // ---- producers, first half: this wave's share of tile t + 1's records (DMA'd three tiles ago) -> the other X tile
            // and the FM terms — eleven events stand behind those DMAs in the wave's queue (waves 6 / 7: the head's store, the
            // meta DMA, tile t + 2's nine record DMAs; waves 4 / 5: the meta DMA and ten DMAs) —, then waves 6 / 7 finish tile
            // t - 1 (head; first trip: count 0, all lanes to the sink)
            ir_wait_vm<11>();
            IR_MARK(4)
            convert((t + 1) & 1, ring_add(s6, 1), XT0 + ((t + 1) & 1) * kIrXT);
            IR_MARK(5)
            if (wave >= 6) head(dA, s6 == 0 ? kIrSlots - 1 : s6 - 1);
            IR_MARK(1)
        }
        ir_barrier();                                       // B: H1 complete; H2 (tile t - 1) consumed
        IR_MARK(2)
        if (wave < 4) {
            // ---- consumers: layer 2 of tile t, output block (mb2, nb2) -> fp32 H2 tile
            f32x16 acc2;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4*>(b2s + nb2 * 32 + 4 * h + 8 * g);
                acc2[4 * g + 0] = bv.x; acc2[4 * g + 1] = bv.y; acc2[4 * g + 2] = bv.z; acc2[4 * g + 3] = bv.w;
            }
            const int row = mb2 * 32 + i32;
            const char* const h1r = H1T + row * (kIrTH * 2);
            bf16x8 af[4];                                   // three steps ahead: this chain's MFMAs are dependent, its reads are not
#pragma unroll
            for (int i = 0; i < 3; ++i) af[i] = *reinterpret_cast<const bf16x8*>(h1r + (((i * 2 + h) ^ sw) << 4));
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                if (ks + 3 < KS2) af[(ks + 3) & 3] = *reinterpret_cast<const bf16x8*>(h1r + ((((ks + 3) * 2 + h) ^ sw) << 4));
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2r[ks], af[ks & 3], acc2, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(H2T + row * kIrHS + nb2 * 32 + 8 * g + 4 * h) =
                    make_float4(acc2[4 * g + 0], acc2[4 * g + 1], acc2[4 * g + 2], acc2[4 * g + 3]);
            IR_MARK(3)
        } else {
            // ---- producers, second half.  The meta slot written a trip ago (ten events behind it: tile t + 2's issue group
            // and, for waves 6 / 7, this trip's store) gives this lane's rows of tile t + 3 and tile t + 4's descriptor; the next
            // meta DMA goes out, then tile t + 3's records into the raw slot just converted (its reads were this wave's own and
            // have returned: the conversion consumed them)
            ir_wait_vm<10>();
            IR_MARK(3)
            uint32_t l_ = (uint32_t)lane;
            asm volatile("" : "+v"(l_));
            const uint32_t* const mb = meta + (pw * 2 + ((t + 1) & 1)) * 64;
            const uint32_t rowA = mb[(l_ >> 3) * 4], rowB = mb[32 + (l_ >> 3) * 4], rowC = mb[l_ & ~3u];
            IrTile d4{(uint32_t)__builtin_amdgcn_readfirstlane((int)mb[1]), (uint32_t)__builtin_amdgcn_readfirstlane((int)mb[2]),
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)mb[3])};
            if (t + 4 >= t_end) d4.cnt = 0;
            meta_issue(t + 5, d4, t & 1);
            issue(d3.req, rowA, rowB, rowC, (t + 1) & 1, ring_add(s6, 3));
            IR_MARK(6)
            // wave 4: the FM prefix of tile t + 2 (DMA'd a trip ago) must have landed before barrier A, behind which every
            // producer reads it.  Behind it in this wave's queue: the 9 record DMAs of its own issue group, this trip's meta DMA
            // and the 10 just issued
            if (wave == 4) ir_wait_vm<20>();
            dA = dB; dB = dC; dC = dD; dD = d3; d3 = d4;
        }
        s6 = ring_add(s6, 1);
    }
#ifdef PG_IR_PROFILE
    if (lane == 0 && blockIdx.x >= 100 && blockIdx.x < 104) {
        uint64_t* o = reinterpret_cast<uint64_t*>(a.sink + 512) + ((blockIdx.x - 100) * 8 + wave) * 8;
        for (int i = 0; i < 8; ++i) o[i] = ph[i];
    }
#endif
    // the head of the last tile (its H2 tile was written in the last phase 2)
    ir_barrier();
    if (wave >= 6) head(dA, s6 == 0 ? kIrSlots - 1 : s6 - 1);
    ir_wait_vm<0>();                                        // (DMAs of tiles past the range are still landing in this workgroup's LDS)
}

bool fm2t_irs_shape(uint32_t th, uint32_t to, uint32_t k, uint32_t nif, int prec) { return prec == 1 && th == 256 && to == 64 && k == 16 && nif == 8; }

int launch_fm2t_irs(pg_ctx* ctx, const MlpArgs& a) {
    constexpr size_t lds = ir_lds_bytes();
    int rc;
    if ((rc = ensure_dyn_lds(ctx, (const void*)fm2t_irs_kernel, lds))) return rc;
    fm2t_irs_kernel<<<ctx->num_cus, 512, lds, ctx->stream>>>(a);
#ifdef PG_IR_PROFILE
    // developer aid (make WS_EXTRA=-DPG_IR_PROFILE): mean cycles per phase of workgroups 100..103, per wave, printed once
    static int calls = 0;
    if (++calls == 8) {
        uint64_t hc[4 * 8 * 8];
        (void)hipMemcpy(hc, a.sink + 512, sizeof hc, hipMemcpyDeviceToHost);
        for (int w = 0; w < 8; ++w) {
            fprintf(stderr, "irs wave %d:", w);
            for (int i = 0; i < 8; ++i) {
                double v = 0;
                for (int g = 0; g < 4; ++g) v += (double)hc[(g * 8 + w) * 8 + i] / 4.0;
                fprintf(stderr, " %9.0f", v);
            }
            fprintf(stderr, "   (barrier A | layer 1 / head | barrier B | layer 2 / rows | wait | convert | issue | loop top)\n");
        }
    }
#endif
    return PG_OK;
}

}  // namespace pg
