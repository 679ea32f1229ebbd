// The body of lattice_wd.hip: everything inside its kernel namespace, written against two macros so that the file can be
// compiled twice into one library --
//   RNNT_WD_NS   the namespace (wd8 / wd16)
//   RNNT_WD_KK   diagonals per block = per interval = per s_barrier
// Blocks of 16 diagonals halve what a block costs besides its diagonals (the barrier, the wait in front of it, the loop
// around it, the loader's and the storer's fixed parts) and pay with twice the predicated work at either end of a column
// block's life and twice the LDS, and a hand-over distance of five intervals that grows with them.  Shipped: 16 from launch
// bound T >= 1024 on -- from T >= 320 where one column block is the whole lattice --, 8 below (lattice_wd.hip:
// wd_block_diagonals has the history).  No include guard: included once per
// instantiation.  Read lattice_wd.hip's header first.
namespace RNNT_WD_NS {

constexpr int K = RNNT_WD_KK;          // diagonals per block = per interval = per s_barrier (8 or 16; lattice_ws.hip: 8)
using ws::f32x2;
using ws::block_barrier;
using ws::compute_block_ip;
using ws::RSRC_WORD3;
using ws::OOB;
constexpr int TRASH = WAVE + K;

#ifndef RNNT_WD_DLOAD
#define RNNT_WD_DLOAD 2
#endif
constexpr int DLOAD = RNNT_WD_DLOAD;   // the loader fetches a block this many intervals before the compute wave reads it
constexpr int PSLOTS = DLOAD + 4;      // LDS ring of pair blocks: fetched DLOAD intervals before the compute wave's first
                                       // read, kept until the storer has taken the last column's label log-probs (3 later)
constexpr int VSLOTS = 2;              // LDS ring of value blocks

constexpr int MIN_SLOTS = 8;           // ... of incoming ones (granules as the neighbour published them); >= DLOAD + 2
static_assert(DLOAD >= 2 && DLOAD + 2 <= MIN_SLOTS, "ring depths");
#ifndef RNNT_WD_SPIN_LIMIT
#define RNNT_WD_SPIN_LIMIT (1 << 21)
#endif
constexpr int SPIN_LIMIT = RNNT_WD_SPIN_LIMIT;   // polls before a hand-over is declared lost (seconds)
#ifndef RNNT_WD_LAG
#define RNNT_WD_LAG 1
#endif
#ifndef RNNT_WD_FAST_TAIL
#define RNNT_WD_FAST_TAIL 1    // the blocks lanes finish in run the hand-written steady-state code (sweep(): full_end)
#endif
#ifndef RNNT_WL_PAD
#define RNNT_WL_PAD 1          // two column blocks: eight waves, the compute waves alone on their SIMDs (k_lattice_wl)
#endif
#ifndef RNNT_WL_PRIO
#define RNNT_WL_PRIO 2         // s_setprio of the compute waves of k_lattice_wl from three column blocks on, where they
#endif                         // share SIMDs with loaders and storers (N=16, T=1500, U=300: 142 -> 137 us; nothing at two)
#ifndef RNNT_WL_DEFAULT_MAX_BLOCKS
#define RNNT_WL_DEFAULT_MAX_BLOCKS 5
#endif
constexpr int LAG = RNNT_WD_LAG; // blocks a column block lets its left neighbour get ahead once it has caught up with it

typedef unsigned long long u64;
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct alignas(16) Smem {
    f32x2 pairs[PSLOTS][K][WAVE];     // [slot][diagonal][position]: the LDS-DMA's landing zone (lane-linear)
    float vals[VSLOTS][K][WAVE];
    u64 mail_raw[MIN_SLOTS][K];       // left neighbour's granules for a block (diagonals d0-1 .. d0+K-2) as they landed
    float mail_vals[MIN_SLOTS][K];    // ... their values, once the loader has checked the tags
    float trash[TRASH];
};

__device__ __forceinline__ unsigned ring_tag(unsigned epoch, unsigned ring) { return epoch ^ (ring * 0x85EBCA6Bu); }
__device__ __forceinline__ unsigned diag_tag(unsigned ring_epoch, int d) {
    const unsigned t = ring_epoch ^ ((unsigned)(d + 1) * 0x9E3779B1u);
    return t ? t : 1u;
}

struct Item { int n, dir, cb; };

__host__ __device__ inline size_t trace_slots(int T, int U) { return (size_t)(T + U - 1) / K + 24; }

// Granules per ring.  The granule of diagonal d sits at index d + 1, so that the eight a consumer block needs
// (diagonals m*K-1 .. m*K+K-2) are one aligned 64-byte group; the last index is a pad that publications of blocks which
// do not exist go to.
__host__ __device__ inline size_t ring_pitch(int T, int U) { return ((size_t)(T + U - 1 + K - 1) / K + 2) * K; }

// 128-bit buffer descriptor in SGPRs for the inline-assembly LDS-DMA (raw buffer, 32-bit data format)
__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes) {
    const u64 a = (u64)p;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = __builtin_amdgcn_readfirstlane(RSRC_WORD3);
    return r;
}
// One LDS-DMA piece: lane l's 16 bytes at (descriptor base + voff + soff) land at LDS byte address lds + 16 l; a voff
// beyond the descriptor's range lands zeros.  M0 (the LDS base) is written in the statement that reads it and left
// there: nothing the compiler generates for these waves reads M0.
__device__ __forceinline__ void dma16(const int voff, const i32x4 rs, const int soff, unsigned lds) {
    lds = __builtin_amdgcn_readfirstlane(lds);
#ifdef RNNT_WD_DMA_ALL_SC1
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen sc1 lds" ::"v"(voff), "s"(rs), "s"(soff), "s"(lds) : "memory");
#else
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rs), "s"(soff), "s"(lds) : "memory");
#endif
}
__device__ __forceinline__ void dma16_agent(const int voff, const i32x4 rs, unsigned lds) {   // agent scope (sc1)
    lds = __builtin_amdgcn_readfirstlane(lds);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen sc1 lds" ::"v"(voff), "s"(rs), "s"(lds) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) const char*)p);
}

// LOCAL: the single-workgroup form (k_lattice_wl below).  All column blocks of the sweep are waves of THIS workgroup and
// meet at every barrier, column block idx running L_LOCAL intervals behind its left neighbour; the boundary column
// goes from the left block's storer straight into the right block's `mail_vals` (sm_right) -- no ring, no tag, no poll,
// nothing in global memory, no preparation launch in front and no redo launch behind.  Everything else (the loader's
// LDS-DMA, the in-place compute blocks, the store-only storer, the dry run) is the code of the distributed form.
constexpr int L_LOCAL = 3;   // a boundary value computed in interval g is stored to the neighbour's LDS by the storer in
                             // g + 1 and read (in place, one block ahead) by the neighbour's compute wave in g + 2
template <bool BETA, bool COMPACT, bool HAS_LEFT, bool HAS_RIGHT, bool LOCAL>
__device__ __forceinline__ void sweep(const LatticeArgs& a, const Item it, const UttLens len, const int nA, Smem& sm,
                                      Smem* sm_right, int* wg_bad, const int role /* 0 compute, 1 loader, 2 storer */) {
    const int n = it.n, idx = it.cb;
    const int Tn = len.Tn, Un = len.Un;
    const int T = COMPACT ? Tn : a.T, U = COMPACT ? Un : a.U;
    const int lane = threadIdx.x & (WAVE - 1);
    const size_t nbase = COMPACT ? compact_base(a, n) : (size_t)n * T * U;
    float* out = (BETA ? a.betas : a.alphas) + nbase;
    if (Un == 1) {   // no labels: prefix / suffix sums by one wave of the first column block's workgroup
        if (idx == 0 && role == 0) {
            const float2* lp2 = reinterpret_cast<const float2*>(a.lp) + nbase;
            const float total = single_column_scan<BETA>(Tn, out, U, lane, [&](int t) { return lp2[(size_t)t * U].x; });
            if (!BETA && lane == 0) a.ll[n] = total;
        }
        return;
    }
    const int ndiag = Tn + Un - 1;
    const float NEG_INF = -__builtin_inff();

    const int wave_c = WAVE * idx;                    // first sweep column of this column block
    const int ucol = wave_c + lane;                   // column in sweep coordinates
    const bool colvalid = ucol < Un;
    const int u = BETA ? (Un - 1 - ucol) : ucol;
    const int uc = min(max(u, 0), U - 1);
    const int ucol_chk = colvalid ? ucol : 0x40000000;
    const int nwa = (Un + WAVE - 1) / WAVE;           // column blocks with a live column
    const int lo = wave_c / K;
    const int hi = (min(ndiag, Tn + wave_c + WAVE) + K - 1) / K;
    if (idx >= nwa || lo >= hi) return;               // nothing to sweep here (uniform)
    // blocks for which the left neighbour publishes a boundary column this block still needs
    const int hi_left = HAS_LEFT ? (min(ndiag, Tn + wave_c) + K - 1) / K : 0;
    // Local time p: one interval per block of K diagonals, one s_barrier per interval, the same [p0, p1) for every wave.
    //   loader   interval p: pairs(p) and the neighbour's block p have landed (checked); fetches both for p+DLOAD
    //   compute  interval p: block p-2 (reads pairs(p-1) for the block after it as it goes)
    //   storer   interval p: values and boundary column of block p-3
    const int p0 = lo - DLOAD, p1 = hi + 3;
    // the window of memory columns the pairs of this column block come from, and where a lane's column sits in it
    const int cwin = BETA ? max(0, Un - WAVE - wave_c) : wave_c;
    const int pos = BETA ? max(0, Un - 1 - ucol - cwin) : lane;
    // hand-over rings in global memory: one per (sweep, column-block boundary)
    const size_t sweep_id = (size_t)2 * n + (BETA ? 1 : 0);
    const size_t pitch = ring_pitch(a.T, a.U);
    u64* ring_in = HAS_LEFT && !LOCAL ? a.mail + (sweep_id * (nA - 1) + (idx - 1)) * pitch : nullptr;
    u64* ring_out = HAS_RIGHT && !LOCAL ? a.mail + (sweep_id * (nA - 1) + idx) * pitch : nullptr;
    const unsigned tag_in = ring_tag(a.epoch, (unsigned)(sweep_id * (nA - 1) + (idx - 1)));
    const unsigned tag_out = ring_tag(a.epoch, (unsigned)(sweep_id * (nA - 1) + idx));
    (void)ring_in; (void)ring_out; (void)tag_in; (void)tag_out;
    // row (forward diagonal mod T) of the first diagonal of block `lo`
    const int dF0 = BETA ? (ndiag - 1 - lo * K) : lo * K;
    const int row0 = ((dF0 % T) + T) % T;
#ifdef RNNT_WD_STATS       // diagnostics build (tools/wd_trace.py): a (sweep, column block, interval) table of s_memrealtime
                           // stamps behind the rings, 8 words per interval: 0/1 compute wave enters / leaves the block of
                           // the interval, 2/3 loader enters / leaves its step, 4 ticks it waited for the neighbour,
                           // 5/6 storer enters / leaves.  Costs a few per cent; never part of the product build.
    u64* const trace = a.mail + (size_t)(gridDim.x / nA) * (nA - 1) * ring_pitch(a.T, a.U) +
                       (((size_t)2 * it.n + it.dir) * nA + idx) * (size_t)trace_slots(a.T, a.U) * 8;
#define RNNT_WD_STAMP(p_, word) do { if (lane == 0) trace[8 * ((p_) + 8) + (word)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RNNT_WD_STAMP(p_, word) do { } while (0)
#endif

    if (role == 1) {
        // ------------------------------ loader wave ------------------------------
        const i32x4 rs_lp = make_rsrc(reinterpret_cast<const float2*>(a.lp) + nbase, (unsigned)((size_t)T * U * 8));
        constexpr bool RING_IN = HAS_LEFT && !LOCAL;           // the neighbour's column arrives through an L2 ring
        const i32x4 rs_ring = make_rsrc(ring_in, RING_IN ? (unsigned)(pitch * 8) : 0u);
        const unsigned lds_pairs = lds_addr(&sm.pairs[0][0][0]);
        const unsigned lds_raw = lds_addr(&sm.mail_raw[0][0]);
        const int rowb = U * 8;
        const int half = lane >> 5;                            // lanes 0-31 fetch diagonal 2j of a block, 32-63 diagonal 2j+1
        const int colb = (cwin + 2 * (lane & 31)) * 8;         // 16 bytes = two columns
        int row_ld = row0;                                     // row of the first diagonal of the next block to fetch
        int slot_ld = 0;                                       // its slot in the LDS ring (block lo = slot 0)
        constexpr int NDMA = K / 2 + (RING_IN ? 1 : 0);        // pieces per interval, always all of them
        // per-lane part of a piece's offset while the K rows of a block do not wrap around the plane (the rule): column
        // + this lane's diagonal relative to the block's lowest row, which goes into the scalar offset
        int voff_j[K / 2];
#pragma unroll
        for (int j = 0; j < K / 2; ++j) voff_j[j] = colb + (BETA ? (K - 1) - (2 * j + half) : (2 * j + half)) * rowb;
        // a poll of the neighbour's block m: lanes 0..K-1 load their granule, load + wait in ONE piece of assembly
        // that leaves the queue empty
        const int mlane = lane < K ? lane : K - 1;
        auto mail_poll = [&](const int m) {
            u64 g;
            asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g) : "v"(ring_in + (m * K + mlane)) : "memory");
            return g;
        };
        auto mail_valid = [&](const int m, const u64 g) {
            const bool ok = (unsigned)(g >> 32) == diag_tag(tag_in, m * K - 1 + mlane);
            return __builtin_amdgcn_ballot_w64(!ok) == 0;
        };
        bool lost = false;     // a wait has timed out: the sweep is flagged for the kernel behind, the rest of it
                               // runs on whatever the ring holds without waiting again
        auto mail_wait = [&](const int m) {                // poll until block m of the neighbour is there
            for (int spins = 0;; ++spins) {
                const u64 g = mail_poll(m);
                if (lost || mail_valid(m, g)) return g;
                if (spins >= SPIN_LIMIT) { atomicOr(wg_bad, 2); lost = true; return g; }
                __builtin_amdgcn_s_sleep(2);
            }
        };
        for (int p = p0; p < p1; ++p) {
            RNNT_WD_STAMP(p, 2);
            wait_vmcnt<NDMA * (DLOAD - 1)>();                  // everything fetched DLOAD intervals ago has landed
            // The neighbour's granules of block p, as they landed: read now, looked at behind this interval's fetches (the
            // LDS round trip in the shadow of their issue; the loader's interval must stay below the compute wave's).
            const bool chk = RING_IN && p >= lo && p < hi_left;
            u64 g = 0;
            if constexpr (RING_IN) g = sm.mail_raw[p & (MIN_SLOTS - 1)][mlane];
            {
                const int pl = p + DLOAD;
                const bool live = pl >= lo && pl < hi;
                const unsigned dst = lds_pairs + (unsigned)slot_ld * (unsigned)(K * WAVE * 8);
                if (BETA ? row_ld >= K - 1 : row_ld + K <= T) {
                    const int soff = (BETA ? row_ld - (K - 1) : row_ld) * rowb;
#pragma unroll
                    for (int j = 0; j < K / 2; ++j) dma16(live ? voff_j[j] : OOB, rs_lp, soff, dst + j * (2 * WAVE * 8));
                } else {                                       // the block's rows wrap (once per sweep; T < K: several times)
#pragma unroll
                    for (int j = 0; j < K / 2; ++j) {
                        int r = BETA ? row_ld - (2 * j + half) : row_ld + (2 * j + half);
                        r = ((r % T) + T) % T;
                        dma16(live ? r * rowb + colb : OOB, rs_lp, 0, dst + j * (2 * WAVE * 8));
                    }
                }
                if (live) {
                    row_ld = BETA ? row_ld - K : row_ld + K;
                    if (T >= K) { if (BETA) { if (row_ld < 0) row_ld += T; } else { if (row_ld >= T) row_ld -= T; } }
                    else row_ld = ((row_ld % T) + T) % T;
                    slot_ld = slot_ld + 1 == PSLOTS ? 0 : slot_ld + 1;
                }
            }
            if constexpr (RING_IN) {
                const int ml = p + DLOAD;
                const int mm = (ml >= lo && ml < hi_left) ? ml : lo;   // (always a block of the ring: the piece is
                                                                        //  issued regardless, its bytes not looked at)
                if (lane < K / 2) dma16_agent(mm * (K * 8) + lane * 16, rs_ring, lds_raw + (unsigned)(ml & (MIN_SLOTS - 1)) * (K * 8));
                if (chk) {
                    const int m = p;
                    if (!mail_valid(m, g)) {
                        // fetched DLOAD intervals ago and the producer had not got there: this column block has
                        // caught up with its neighbour.  Let the neighbour get LAG blocks ahead (or finish) before
                        // going on, so that the look-ahead fetches of the following blocks find their data.
#ifdef RNNT_WD_STATS
                        const u64 t_wait = __builtin_amdgcn_s_memrealtime();
#endif
                        if (LAG > 0 && m > lo) mail_wait(min(m + LAG, hi_left - 1));   // (the first block: at once)
                        g = mail_wait(m);
                        // what was fetched ahead for the following blocks was fetched before this one existed: again --
                        // and, these pieces being the youngest in the queue now, waited for here (the counted wait at the
                        // head of the next interval would leave them in flight)
#pragma unroll
                        for (int q = 1; q < DLOAD; ++q) {
                            const int mq = min(m + q, hi_left - 1);
                            if (lane < K / 2) dma16_agent(mq * (K * 8) + lane * 16, rs_ring, lds_raw + (unsigned)((m + q) & (MIN_SLOTS - 1)) * (K * 8));
                        }
                        wait_vmcnt<0>();
#ifdef RNNT_WD_STATS
                        if (lane == 0) trace[8 * (p + 8) + 4] = __builtin_amdgcn_s_memrealtime() - t_wait;
#endif
                    }
                    if (lane < K) sm.mail_vals[m & (MIN_SLOTS - 1)][lane] = __builtin_bit_cast(float, (unsigned)g);
                }
            }
            RNNT_WD_STAMP(p, 3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (its own LDS writes; NOT the pieces in flight)
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    if (role != 1) {
        // ------------------------------ compute wave (and the storer's dry run) ------------------------------
        // The storer wave runs this loop "dry" before it takes up its own work: one block per variant this column block
        // will run, on whatever the registers and LDS hold, no barrier -- for the instruction cache (header, point 3).
        // Everything it writes (vals) is written again by the real blocks before anybody reads it: no real
        // block starts before the storer has reached the first barriers.
        const bool dry = role == 2;
        float Y = (ucol == 0) ? 0.0f : NEG_INF;
        float X = NEG_INF;
        ws::f32x4 cur2[K / 2];                                 // pairs of diagonals 2j, 2j+1: (x, y) and (z, w)
        ws::f32x4 seed4[K / 4];                                // the neighbour's boundary values of diagonals 4j ... 4j+3
        // Blocks [lo, head_end) have lanes that start inside them (head variant: the rim select only), [full_end, hi) lanes
        // that finish (general predicated variant), [head_end, full_end) have every lane that owns a column live
        // throughout (lanes beyond the last column run the unpredicated code too: their values only travel right, their
        // stores are dropped).
        // d0 > the last column's first diagonal: that diagonal itself is a rim cell (it takes `emit`, lattice_step.h), and
        // what a lane holds before it is unspecified in the head variant -- it must not fall into a steady-state block
        const int fb0 = max(lo, min(wave_c + WAVE - 1, Un - 1) / K + 1);
        const int fb1 = min(hi, (wave_c + Tn) / K);                              // d0 + K <= the first column's end
        int lb, head_end, full_end, tail_end;
        if (dry) {
            // pseudo blocks: -2 in the head variant (a first column block goes straight into it at launch: warming it
            // here would only delay the first barrier), -1 in the steady-state one (needed 64 / K blocks later)
            lb = idx > 0 ? -2 : -1;
            head_end = -1; full_end = 0; tail_end = 0;
#pragma unroll
            for (int j = 0; j < K / 4; ++j) seed4[j] = ws::f32x4{-3.0f, -3.0f, -3.0f, -3.0f};
#pragma unroll
            for (int j = 0; j < K / 2; ++j) cur2[j] = ws::f32x4{-1.0f, -2.0f, -1.0f, -2.0f};
        } else {
            // block lb is computed during interval lb + 2: its pairs and the neighbour's block have landed and are
            // checked by the end of interval lb
            for (int t = p0; t < lo + 1; ++t) block_barrier();
            const f32x2* src = &sm.pairs[0][0][pos];
#pragma unroll
            for (int j = 0; j < K / 4; ++j) {
                const float* mv = &sm.mail_vals[lo & (MIN_SLOTS - 1)][4 * j];
                seed4[j] = HAS_LEFT ? ws::f32x4{mv[0], mv[1], mv[2], mv[3]} : ws::f32x4{NEG_INF, NEG_INF, NEG_INF, NEG_INF};
            }
#pragma unroll
            for (int j = 0; j < K / 2; ++j) {
                const f32x2 a0 = src[(2 * j) * WAVE], a1 = src[(2 * j + 1) * WAVE];
                cur2[j] = ws::f32x4{a0.x, a0.y, a1.x, a1.y};
            }
            block_barrier();
            lb = lo;
            // (a lattice so short that lanes start and finish in the same blocks: everything in the general variant)
            head_end = fb0 < fb1 ? fb0 : lo; full_end = fb0 < fb1 ? fb1 : lo; tail_end = hi;
#if RNNT_WD_FAST_TAIL
            // The blocks lanes FINISH in (none starts: they lie behind fb0) run the steady-state code as well.  What a lane
            // computes behind its last frame reaches no result: the storer predicates the stores of every block outside
            // [sfb0, sfb1) per lane; its right neighbour's last live cell (row T_n - 1, one diagonal later) reads the lane's
            // own LAST LIVE value, and so does the next column block's lane 0 from the boundary column (column c is live on
            // diagonals [c, c + T_n - 1], column c + 1 reads it on [c, c + T_n - 1]).  The one thing the general variant's
            // per-lane freeze is needed for is alpha's log-likelihood -- Y of the lane that owns column U_n - 1, read behind
            // the loop -- and that lane finishes on the sweep's very last diagonal: the last block of the last column block
            // of an alpha sweep stays in the general variant, nothing else.
            if (fb0 < fb1) full_end = max(fb1, (!BETA && idx == nwa - 1) ? hi - 1 : hi);
#endif
        }
        int slot = 0;                                          // LDS slot of block lb's pairs (block lo = slot 0)
        const unsigned pairs0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)&sm.pairs[0][0][pos];
        const unsigned seeds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)&sm.mail_vals[0][0];
        // what block lb needs besides its registers: the LDS addresses of the NEXT block's pairs and seeds (reloaded in
        // place as it goes) and its own slot of the value ring.  Worked out for the first block here and for every
        // following one at the END of its predecessor, in front of the wait for that block's last reloads -- arithmetic
        // that needs nothing from LDS, in the shadow of an LDS round trip the wave would otherwise sit out.
        unsigned nsrc = 0, nseed = 0;
        ws::lds_float* vslot = nullptr;
        auto prepare_block = [&]() {
            slot = slot + 1 == PSLOTS ? 0 : slot + 1;          // the NEXT block's slot: its pairs landed an interval ago
            nsrc = pairs0 + (unsigned)slot * (unsigned)(K * WAVE * 8);
            nseed = seeds0 + (unsigned)((lb + 1) & (MIN_SLOTS - 1)) * (K * 4);
            vslot = (ws::lds_float*)&sm.vals[lb & (VSLOTS - 1)][0][lane];
        };
        prepare_block();
        auto one_block = [&](auto mode_c) {
            constexpr int MODE = decltype(mode_c)::value;
            const int d0 = lb * K;
            RNNT_WD_STAMP(lb + 2, 0);
            // (no mailbox write on this wave: the storer rebuilds the boundary column from the values)
            compute_block_ip<K, BETA, MODE, HAS_LEFT>(cur2, seed4, nsrc, nseed, Y, X, d0, ucol_chk, Tn, vslot, d0 - wave_c);
            RNNT_WD_STAMP(lb + 2, 1);
            ++lb;
            prepare_block();
            asm volatile("" : "+v"(nsrc), "+s"(nseed), "+v"(vslot));   // (worked out HERE, not behind the barrier)
            // The value stores and the in-place reloads of the block (inline assembly, which no fence of the compiler's counts)
            // have to be complete: one explicit wait, with a "memory" clobber that also keeps the compiler's own LDS
            // accesses on their side of the barrier.  Zero, not a count (lattice_step.h) -- and IN THE DRY RUN TOO.  Until
            // the end of round 5 the storer's dry blocks skipped this wait with the barrier, and went on into the storer's
            // own code with their last reloads still on the way to registers that were dead by then and that the compiler
            // handed out again at once -- to the storer's store offsets, among others.  A reload that landed late (an LDS
            // kept busy by other processes' workgroups) overwrote them: a lane's values stored into another column for
            // the rest of the sweep (zeros from the not yet filled tile = column 0), granules published to addresses
            // nobody polls (a hand-over "lost" after a second of polling).  tools/check_inplace_reloads.py, once it
            // followed every edge of the control-flow graph instead of one path, pointed at it; tools/wd_soak.py had
            // shown the symptoms (profiles/r05_wd_soak.txt: the blocks of 16 diagonals, whose twelve reloads per dry block
            // happened to share registers with the offsets).
            ws::wait_lds_keep<K, HAS_LEFT>(cur2, seed4);   // (the refilled registers are operands of the wait: lattice_step.h)
            if (!dry) {
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        };
        // head (lanes start), steady state, tail (lanes finish; for short lattices: everything): each variant exists once
#pragma nounroll
        while (lb < head_end) one_block(std::integral_constant<int, ws::BLOCK_HEAD>{});
#pragma nounroll
        while (lb < full_end) one_block(std::integral_constant<int, ws::BLOCK_FULL>{});
#pragma nounroll
        while (lb < tail_end) one_block(std::integral_constant<int, ws::BLOCK_MASKED>{});
        if (!dry) {
            for (int t = hi + 2; t < p1; ++t) block_barrier();
            if constexpr (!BETA) {
                // Y of a finished lane is frozen at alpha + lpB of its last live cell (core_gather.cu:339)
                if (ucol == Un - 1) a.ll[n] = Y;
            }
            return;
        }
    }

    // ------------------------------ storer wave ------------------------------
    // One interval of this wave must not take longer than one of the compute wave's (0.32 - 0.36 us since round 5: the
    // barrier makes the slowest wave everybody's pace).  So: every LDS read of the interval is issued up front (one
    // exposed LDS latency, not one per consumer), the steady-state blocks -- every lane that owns a column live, the K
    // rows not wrapping around the plane -- store without per-lane predicates and with one scalar add per row, and only
    // the blocks at either end of the column block's life (and the one block per sweep whose rows wrap) take the general
    // form.
    {
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, T * U * 4, RSRC_WORD3);
        const int voff_out = colvalid ? uc * 4 : OOB;
        const int rowb = U * 4;
        int row_st = row0;
        int slot_st = 0;                                       // LDS slot of block ps's pairs (block lo = slot 0)
        // (the compute wave's fb0 / fb1: blocks [sfb0, sfb1) have no lane that starts or finishes inside them)
        const int sfb0 = max(lo, min(wave_c + WAVE - 1, Un - 1) / K + 1);
        const int sfb1 = min(hi, (wave_c + Tn) / K);
        const int plane = lane < K ? lane : K - 1;             // the lane's diagonal of the boundary column
        for (int p = p0; p < p1; ++p) {
            const int ps = p - 3;
            const bool ps_live = ps >= lo && ps < hi;
            RNNT_WD_STAMP(p, 5);
            // what the compute wave's lane 63 handed to its DPP shift after diagonal ps * K + plane: its value (beta), its
            // value + the label log-prob of its cell (alpha) -- the same fp32 addition, the same bits.  (Where lane 63 is
            // not live the result is meaningless and no live cell of the neighbour reads it.)
            float x = 0.0f, xl = 0.0f;
            if constexpr (HAS_RIGHT) {
                x = sm.vals[ps & (VSLOTS - 1)][plane][WAVE - 1];
                if constexpr (!BETA) xl = sm.pairs[slot_st][plane][WAVE - 1].y;   // (still in the ring: PSLOTS)
            }
            float v[K];
            const float* src = &sm.vals[ps & (VSLOTS - 1)][0][lane];
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = src[k * WAVE];
            if constexpr (HAS_RIGHT) {
                if constexpr (!BETA) x += xl;
                if constexpr (LOCAL) {
                    // straight into the right neighbour's seeds: diagonal d is seed (d + 1) mod K of its block (d + 1) / K
                    // (the neighbour reads block m's seeds two intervals from now: L_LOCAL)
                    if (lane < K && ps_live) {
                        const int at = ps * K + lane + 1;
                        sm_right->mail_vals[(at / K) & (MIN_SLOTS - 1)][at % K] = x;
                    }
                } else {
                    // (a block that does not exist goes to the ring's pad granule, which nobody reads; K lanes store --
                    // an agent-scope store is one fabric write per lane)
                    if (lane < K) {
                        const int d = ps * K + lane;
                        const size_t at = ps_live ? (size_t)(d + 1) : pitch - 1;
                        const u64 g = ((u64)diag_tag(tag_out, d) << 32) | __builtin_bit_cast(unsigned, x);
                        __hip_atomic_store(ring_out + at, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            const bool nowrap = BETA ? row_st >= K - 1 : row_st + K <= T;
            if (ps >= sfb0 && ps < sfb1 && nowrap) {
                int soff = row_st * rowb;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v[k]), rs_out, voff_out, soff, 0);
                    soff = BETA ? soff - rowb : soff + rowb;
                }
                row_st = BETA ? row_st - K : row_st + K;
                if (BETA) { if (row_st < 0) row_st += T; } else { if (row_st >= T) row_st -= T; }
                slot_st = slot_st + 1 == PSLOTS ? 0 : slot_st + 1;
            } else {
                const int d0 = ps * K;
                const int vo = ps_live ? voff_out : OOB;
                int r = row_st;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const bool live = (unsigned)(d0 + k - ucol_chk) < (unsigned)Tn;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v[k]), rs_out, live ? vo : OOB, r * rowb, 0);
                    r = BETA ? (r == 0 ? T - 1 : r - 1) : (r + 1 == T ? 0 : r + 1);
                }
                if (ps_live) { row_st = r; slot_st = slot_st + 1 == PSLOTS ? 0 : slot_st + 1; }
            }
            RNNT_WD_STAMP(p, 6);
            if constexpr (HAS_RIGHT && LOCAL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the seeds it wrote
            __builtin_amdgcn_s_barrier();                      // (its LDS reads are complete: the stores needed them)
        }
    }
}

template <bool COMPACT>
__global__ void __launch_bounds__(3 * WAVE) k_lattice_wd(LatticeArgs a, const int nA) {
    __shared__ Smem sm;
    __shared__ int wg_bad, s_item;
    // launch epoch = host counter (constant across the replays of a captured graph) + the library's per-device launch
    // counter (queue[1], bumped by the preparation kernel in front of every launch, replayed or not)
    // (a.queue == nullptr: the launch of single-column-block lattices -- nothing is handed over, so there is no work
    //  queue, no ring, no tag and nothing that could flag a sweep: items in launch order)
    if (a.queue) a.epoch += (unsigned)a.queue[1];
    if (threadIdx.x == 0) { s_item = a.queue ? atomicAdd(a.queue, 1) : (int)blockIdx.x; wg_bad = 0; }
    __syncthreads();
    const int sweeps = gridDim.x / nA;                 // 2N
    Item it;
    it.cb = s_item / sweeps;
    const int s = s_item - it.cb * sweeps;
    it.n = s >> 1;
    it.dir = s & 1;
    const UttLens len = utt_lens<COMPACT>(a.xn, a.yn, it.n, a.T, a.U);
    // (compact: an utterance with bad lengths has no plane of its own to sweep; beta_only: the alpha plane is not the
    //  caller's to write -- run_warp_rnnt_compact with required_grad = false)
    if ((!COMPACT || len.ok) && !(a.beta_only && !it.dir)) {
        const bool hl = it.cb > 0, hr = it.cb + 1 < (len.Un + WAVE - 1) / WAVE;
#define RNNT_WD_SWEEP(B)                                                                    \
    do {                                                                                    \
        if (hl) { if (hr) sweep<B, COMPACT, true, true, false>(a, it, len, nA, sm, nullptr, &wg_bad, role);       \
                  else sweep<B, COMPACT, true, false, false>(a, it, len, nA, sm, nullptr, &wg_bad, role); }       \
        else { if (hr) sweep<B, COMPACT, false, true, false>(a, it, len, nA, sm, nullptr, &wg_bad, role);         \
               else sweep<B, COMPACT, false, false, false>(a, it, len, nA, sm, nullptr, &wg_bad, role); }         \
    } while (0)
        const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        if (it.dir) RNNT_WD_SWEEP(true); else RNNT_WD_SWEEP(false);
#undef RNNT_WD_SWEEP
    }
    __syncthreads();
    // (Round 6 tried to redo a flagged sweep HERE, by the last of the sweep's workgroups to finish, instead of by a kernel
    //  launched behind this one on every call: every workgroup then has to release its plane stores at agent scope before
    //  it counts itself done -- a write-back of its XCD's L2 -- and that costs more than the idle launch it saves: alpha+beta
    //  alone 105.0 vs 101.8 us at N=16, T=1500, U=300 and 80.3 vs 74.0 at N=32, T=1000, U=200 with the 5 us launch counted
    //  on the other side; profiles/r06_lattice_helpers_ab.txt.  Removed.)
    if (threadIdx.x == 0 && wg_bad && a.redo) atomicOr(&a.redo[2 * it.n + it.dir], wg_bad);
}

// ---------------------------------------------------------------------------------------------------------------
// k_lattice_wl: the same sweep with ALL column blocks of a sweep in one workgroup (three waves each), for lattices that
// are too short, or batches that are too large, for a workgroup per column block to pay for its ring preparation, its
// redo launch and its hand-over through L2 -- the place of lattice_ws.hip, with this file's wave roles (round 5).
// Global interval g: column block idx runs its local time p = g - L_LOCAL * idx; every wave executes the same
// g_end - g_begin barriers (idle ones in front of its block's first interval and behind its last).
// ---------------------------------------------------------------------------------------------------------------
template <bool COMPACT, int NA_MAX>
__global__ void __launch_bounds__((NA_MAX == 2 ? 8 : 3 * NA_MAX) * WAVE) k_lattice_wl(const LatticeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char wl_smem[];
    Smem* const sms = reinterpret_cast<Smem*>(wl_smem);
    // XCD-aware placement as in lattice_ws.hip: the alpha and the beta sweep of an utterance on one XCD (speed only)
    const unsigned b = blockIdx.x, pairs_total = gridDim.x >> 1;
    const unsigned grp = b >> 4, in = b & 15;
    unsigned n, dir;
    if ((grp << 3) + 8 <= pairs_total) { n = (grp << 3) + (in & 7); dir = in >> 3; }
    else { const unsigned r = b - (grp << 4); n = (grp << 3) + (r >> 1); dir = r & 1; }   // tail group
    if (a.beta_only && !dir) return;
    if (a.redo && a.redo[2 * n + dir] == 0) return;   // launched behind a ring kernel: only the sweeps it flagged
    const UttLens len = utt_lens<COMPACT>(a.xn, a.yn, (int)n, a.T, a.U);
    if (COMPACT && !len.ok) return;                    // no plane of its own to sweep (uniform)
    // Which wave does what.  A workgroup's waves go to the CU's four SIMDs in a fixed cyclic order, so waves w and w + 4
    // share one.  A compute wave that shares its SIMD with a loader or a storer loses issue slots to it -- and the sweep
    // runs at the pace of its slowest compute wave (two column blocks, six waves in column-block-major order: 57 ns per
    // diagonal against 47 for a compute wave alone on its SIMD).  With two column blocks the workgroup is launched with
    // eight waves: [compute 0, compute 1, loader 0, storer 0, -, -, loader 1, storer 1]; the two spare waves end at once
    // (ended waves do not take part in barriers), the compute waves keep a SIMD each and the four helpers share the
    // other two.  From three column blocks on the compute waves cannot all be alone: column-block-major order.
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int idx, role, nA;
    if (RNNT_WL_PAD && NA_MAX == 2 && blockDim.x == 8 * WAVE) {
        if (w == 4 || w == 5) return;
        nA = 2;
        idx = (w == 1 || w >= 6) ? 1 : 0;
        role = w < 2 ? 0 : ((w == 2 || w == 6) ? 1 : 2);
    } else {
        idx = w / 3; role = w - 3 * idx;
        nA = blockDim.x / (3 * WAVE);
    }
    if (RNNT_WL_PRIO && role == 0 && nA >= 3) __builtin_amdgcn_s_setprio(RNNT_WL_PRIO);   // (the dependent chain first)
    Item it;
    it.n = (int)n; it.dir = (int)dir; it.cb = idx;
    if (len.Un == 1) {                                 // no labels: one wave's prefix / suffix sums (uniform, no barrier)
        if (w == 0) {
            if (dir) sweep<true, COMPACT, false, false, true>(a, it, len, nA, sms[0], nullptr, nullptr, 0);
            else sweep<false, COMPACT, false, false, true>(a, it, len, nA, sms[0], nullptr, nullptr, 0);
        }
        return;
    }
    const int Tn = len.Tn, Un = len.Un, ndiag = Tn + Un - 1;
    const int nwa = (Un + WAVE - 1) / WAVE;            // column blocks with a live column
    auto lo_of = [&](int i) { return WAVE * i / K; };
    auto hi_of = [&](int i) { return (min(ndiag, Tn + WAVE * i + WAVE) + K - 1) / K; };
    const int g_begin = lo_of(0) - DLOAD;
    const int g_end = hi_of(nwa - 1) + 3 + L_LOCAL * (nwa - 1);
    if (idx >= nwa) {                                  // a padded batch: this utterance is narrower than the launch
        for (int g = g_begin; g < g_end; ++g) __builtin_amdgcn_s_barrier();
        return;
    }
    const int g0 = lo_of(idx) - DLOAD + L_LOCAL * idx, g1 = hi_of(idx) + 3 + L_LOCAL * idx;
    for (int g = g_begin; g < g0; ++g) __builtin_amdgcn_s_barrier();
    const bool hl = idx > 0, hr = idx + 1 < nwa;
    Smem& sm = sms[idx];
    Smem* const smr = hr ? &sms[idx + 1] : nullptr;
#define RNNT_WL_SWEEP(B)                                                                                     \
    do {                                                                                                     \
        if (hl) { if (hr) sweep<B, COMPACT, true, true, true>(a, it, len, nA, sm, smr, nullptr, role);       \
                  else sweep<B, COMPACT, true, false, true>(a, it, len, nA, sm, smr, nullptr, role); }       \
        else { if (hr) sweep<B, COMPACT, false, true, true>(a, it, len, nA, sm, smr, nullptr, role);         \
               else sweep<B, COMPACT, false, false, true>(a, it, len, nA, sm, smr, nullptr, role); }         \
    } while (0)
    if (dir) RNNT_WL_SWEEP(true); else RNNT_WL_SWEEP(false);
#undef RNNT_WL_SWEEP
    for (int g = g1; g < g_end; ++g) __builtin_amdgcn_s_barrier();
}

}  // namespace RNNT_WD_NS
