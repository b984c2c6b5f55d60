// compose.hip — composition (+ trim, + fused n=1 shortest path) of HBM-resident CSR FSTs.
//
// Replaces, for SortedMatcher x2 + SequenceComposeFilter (the AutoFilter default):
//   compose / compose_with_config      rustfst/src/algorithms/compose/compose_static.rs:166-306
//   ComposeFstOp::{compute_start,compute_trs,compute_final_weight,match_input,ordered_expand,
//                  match_tr,match_tr_selected,add_tr}   compose/compose_fst_op.rs:199-449
//   SortedMatcher / IteratorSortedMatcher / eps_loop   compose/matchers/sorted_matcher.rs:44-184, matchers/mod.rs:98-105
//   SequenceComposeFilter::{set_state,filter_tr}       compose/compose_filters/sequence_compose_filter.rs:134-171
//   StateTable::find_id                                 lazy/state_table.rs:49-59,102-119
//   LazyFst::compute (FIFO BFS = the state numbering)   lazy/lazy_fst.rs:226-269
//   connect + del_states                                connect.rs:51-66, vector_fst/mutable_fst.rs:132-189
//
// Execution model ("one wavefront = one composition problem"): the reference is a sequential BFS whose
// state ids are first-touch order.  A 64-lane wave walks that BFS level by level for ONE (fst1,fst2)
// pair; a launch runs one wave per problem, so a batch of acceptors against a shared transducer fills
// the chip with independent waves (8 XCDs x 32 CUs x up to 32 waves) and needs no inter-workgroup
// synchronisation at all.  Inside a level the 64 lanes share the work of one composed state:
//   - the searched side's arc block (<= 64 arcs) is read with ONE coalesced 16-B-per-lane load and kept
//     in registers; label matching is a wave ballot, matched arcs are fetched with lane shuffles
//     (no second trip to HBM); larger blocks fall back to a per-lane binary search;
//   - emitted arcs get their slot by a wave prefix sum, destination tuples go through an
//     open-addressing table (u64 key CAS) whose value word holds atomicMin(first emission index);
//   - after the level, first occurrences are ranked with ballots: new ids = reference BFS ids.
// Everything stays in HBM/L2 between phases: the fused pipeline (compose -> relax -> backtrace) never
// returns to the host.
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "compose_filters.h"
#include "fst_props.h"
#include "host_parallel.h"

namespace wfst {

namespace {

constexpr uint32_t NO_LABEL = WFST_NO_LABEL;
constexpr uint32_t REJECT = 0xFFFFFFFFu;  // FilterState::new_no_state()
// bits of the 4th word of a state record (fst_store.hip derive_noeps_kernel; common.h SREC_*)
using wfst::SREC_ALL_IEPS;
using wfst::SREC_ALL_OEPS;
using wfst::SREC_NO_IEPS;
using wfst::SREC_NO_OEPS;
constexpr uint64_t HT_EMPTY = ~0ull;
constexpr uint32_t HV_INIT = 0xFFFFFFFFu;
constexpr uint32_t HV_PEND = 0x80000000u;
constexpr uint64_t KEY_INF = ~0ull;

enum : uint32_t { MODE_BOTH = 0, MODE_INPUT = 1, MODE_OUTPUT = 2 };  // MatchType after match_type()
enum : uint32_t {
  ST_OK = 0, ST_OVERFLOW_STATES = 1, ST_OVERFLOW_ARCS = 2, ST_OVERFLOW_HASH = 3, ST_OVERFLOW_PATH = 4,
  ST_NOT_A_STRING_CASE = 5,  // the string o T kernel met a case it does not cover: redo on the general kernel
  ST_SWITCH_WIDE = 6,        // a frontier too wide for one wave: compose() redoes the pair on the wide driver
  ST_TIE_ORDER = 7           // fused shortest path: a state of the path has no predecessor tight in the hop count (inexact
                             // f32 sums, sssp.hip parent_class): the host redoes the problem as compose + shortest_path
};
enum : uint32_t { FLAG_TRIM = 1, FLAG_SP = 2 };

struct FstView {
  const uint32_t* offsets;
  const wfst_tr* arcs;
  const float* finals;
  const uint32_t* noeps;
  const uint4* srec;   // {arc begin, arc count, final bits, SREC_* epsilon facts}
  const uint2* anext;  // per arc {arc begin, arc count} of its destination state (string o T kernel only), or null
  uint32_t n_states;
  int32_t start;  // -1 = None
  uint64_t n_arcs;
};

struct ProblemDesc {
  FstView f1;
  uint32_t mode;
  uint32_t filter;  // ComposeFilterEnum value (compose_static.rs:19-33); 0 (Auto) == 3 (Sequence)
};

struct Caps {
  uint32_t S;  // composed states
  uint32_t A;  // composed arcs
  uint32_t H;  // hash slots (power of two)
  uint32_t W;  // give up (ST_SWITCH_WIDE) when a BFS level adds more states than this; 0 = never
};

struct Result {
  uint32_t status;
  uint32_t n_states, n_arcs;  // pre-trim
  uint32_t t_states, t_arcs;  // after trim (== pre-trim when not trimming)
  int32_t t_start;
  uint32_t has_path, hops;
  float final_weight, total;
  uint32_t path_off;  // offset of the path arcs in the packed path buffer
  uint32_t n_levels;
  uint32_t facts;     // string_compose_sp_kernel: OR of the path arcs' property facts (fst_props.h: path_arc_facts), or
                      // PATH_FACTS_NONE: the host knows the path FST's properties without reading its arcs back
  uint32_t done;      // string_compose_sp_kernel, results in pinned memory: the run's ticket, written after everything else of
                      // the problem (system-scope fence in between): the host waits for these words instead of the stream
#ifdef WFST_PHASE_TIMING
  unsigned long long dbg[8];
#endif
};

struct Arena {
  uint64_t* tuples;   // [S]   packed (fs<<63 | s1<<32 | s2)
  uint64_t* hkeys;    // [H]
  uint64_t* skey;     // [S]   shortest-path key (enc(d)<<32 | hops)
  uint64_t* parent;   // [S]
  wfst_tr* arcs;      // [A]   composed arcs (pre-trim), nextstate patched level by level
  wfst_tr* t_arcs;    // [A]   trimmed arcs
  uint32_t* hvals;    // [H]
  uint32_t* off;      // [S+1]
  uint32_t* t_off;    // [S+1]
  float* fin;         // [S]
  float* t_fin;       // [S]
  uint32_t* aux;      // [S]   coaccessible flag / mark
  uint32_t* aux2;     // [S]   new id / jump pointer
  uint32_t* lvl;      // [S+1] first id of each BFS level
};

__host__ __device__ inline size_t al16(size_t x) { return (x + 15) & ~(size_t)15; }
__host__ __device__ inline size_t arena_bytes(const Caps& c) {
  size_t b = 0;
  b += al16((size_t)c.S * 8);        // tuples
  b += al16((size_t)c.H * 8);        // hkeys
  b += al16((size_t)c.S * 8);        // skey
  b += al16((size_t)c.S * 8);        // parent
  b += al16((size_t)c.A * 16);       // arcs
  b += al16((size_t)c.A * 16);       // t_arcs
  b += al16((size_t)c.H * 4);        // hvals
  b += al16((size_t)(c.S + 1) * 4);  // off
  b += al16((size_t)(c.S + 1) * 4);  // t_off
  b += al16((size_t)c.S * 4);        // fin
  b += al16((size_t)c.S * 4);        // t_fin
  b += al16((size_t)c.S * 4);        // aux
  b += al16((size_t)c.S * 4);        // aux2
  b += al16((size_t)(c.S + 1) * 4);  // lvl
  return (b + 255) & ~(size_t)255;
}
__device__ inline Arena carve_arena(char* base, const Caps& c) {
  Arena a;
  char* p = base;
  a.tuples = (uint64_t*)p; p += al16((size_t)c.S * 8);
  a.hkeys = (uint64_t*)p; p += al16((size_t)c.H * 8);
  a.skey = (uint64_t*)p; p += al16((size_t)c.S * 8);
  a.parent = (uint64_t*)p; p += al16((size_t)c.S * 8);
  a.arcs = (wfst_tr*)p; p += al16((size_t)c.A * 16);
  a.t_arcs = (wfst_tr*)p; p += al16((size_t)c.A * 16);
  a.hvals = (uint32_t*)p; p += al16((size_t)c.H * 4);
  a.off = (uint32_t*)p; p += al16((size_t)(c.S + 1) * 4);
  a.t_off = (uint32_t*)p; p += al16((size_t)(c.S + 1) * 4);
  a.fin = (float*)p; p += al16((size_t)c.S * 4);
  a.t_fin = (float*)p; p += al16((size_t)c.S * 4);
  a.aux = (uint32_t*)p; p += al16((size_t)c.S * 4);
  a.aux2 = (uint32_t*)p; p += al16((size_t)c.S * 4);
  a.lvl = (uint32_t*)p;
  return a;
}

// ---------------------------------------------------------------- wave helpers (wave = 64 lanes)
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t lanes_below(uint64_t mask) {  // popcount of mask bits below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t lane, uint32_t* total) {
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t y = __shfl_up(x, d);
    if (lane >= (uint32_t)d) x += y;
  }
  *total = __shfl(x, 63);
  return x - v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor(v, d));
  return v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    unsigned long long o = __shfl_xor(v, d);
    v = o < v ? o : v;
  }
  return v;
}
// broadcast from a wave-uniform lane: v_readlane (scalar path) instead of a ds_bpermute round trip
__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t src) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src);
}
__device__ __forceinline__ uint64_t rl64(uint64_t v, uint32_t src) {
  return ((uint64_t)rl((uint32_t)(v >> 32), src) << 32) | rl((uint32_t)v, src);
}
__device__ __forceinline__ uint4 rl128(uint4 v, uint32_t src) {
  return make_uint4(rl(v.x, src), rl(v.y, src), rl(v.z, src), rl(v.w, src));
}
// LDS traffic of one wave is processed in issue order, so lanes of the (single-wave) workgroup only need
// the compiler not to reorder around the hand-off and the LDS counter drained: no vmcnt wait, no s_barrier.
__device__ __forceinline__ void lds_handoff() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// The wave is a whole workgroup (blockDim == 64): __syncthreads() is the wave-level fence that makes
// the lanes' global-memory writes visible to each other between phases.
__device__ __forceinline__ void wave_sync() { __syncthreads(); }

template <class T>
__device__ __forceinline__ T ld_l2(const T* p) {  // L2-served load for words other lanes update atomically
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T>
__device__ __forceinline__ void st_l2(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t enc_f32(float f) {
  uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(uint32_t e) {
  uint32_t b = (e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e;
  return __uint_as_float(b);
}
// TropicalWeight::times_assign (semirings/tropical_weight.rs:60-70)
__device__ __forceinline__ float wtimes(float a, float b) { return a == INF ? a : (b == INF ? b : a + b); }

__device__ __forceinline__ uint32_t hash_u64(uint64_t h) {
  h ^= h >> 33;
  h *= 0xff51afd7ed558ccdull;
  h ^= h >> 33;
  h *= 0xc4ceb9fe1a85ec53ull;
  h ^= h >> 33;
  return (uint32_t)h;
}
// ComposeStateTuple {fs, s1, s2} in one u64: fs (0..2) : 2 | s1 : 30 | s2 : 32
constexpr uint32_t TUPLE_S1_MASK = 0x3FFFFFFFu;
__device__ __forceinline__ uint64_t pack_tuple(uint32_t fs, uint32_t s1, uint32_t s2) {
  return ((uint64_t)fs << 62) | ((uint64_t)s1 << 32) | s2;
}
__device__ __forceinline__ uint32_t tuple_s1(uint64_t tk) { return (uint32_t)(tk >> 32) & TUPLE_S1_MASK; }

// StateTable::find_id (lazy/state_table.rs:49-59) as an open-addressing table.  The value word is
// atomicMin(first emission index of this level) while the tuple is new, then its final id.
__device__ __forceinline__ uint32_t ht_insert(uint64_t* hkeys, uint32_t* hvals, uint32_t hmask, uint64_t key,
                                              uint32_t pend) {
  uint32_t slot = hash_u64(key) & hmask;
  for (uint32_t probes = 0; probes <= hmask; ++probes) {
    const uint64_t k = hkeys[slot];  // may be a stale EMPTY: the CAS below is the arbiter
    if (k == key) break;
    if (k == HT_EMPTY) {
      const uint64_t prev = atomicCAS((unsigned long long*)&hkeys[slot], (unsigned long long)HT_EMPTY,
                                      (unsigned long long)key);
      if (prev == HT_EMPTY || prev == key) break;
    }
    slot = (slot + 1) & hmask;
  }
  atomicMin(&hvals[slot], pend);
  return slot;
}

struct ArcReg {
  uint32_t il, ol;
  float w;
  uint32_t ns;
};
// Pointers that reach the kernel through a descriptor in memory (fst1 of every problem) are generic to the compiler,
// which then emits FLAT loads: slower, and they tie the LDS and the vector-memory wait counters together.  Everything an
// FstView points to is device memory, so say so.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef const u32x4_t __attribute__((address_space(1)))* global_u4_ptr;
__device__ __forceinline__ uint4 ld_global16(const void* p) {
  // (the integer round trip is what makes the compiler drop "generic")
  const u32x4_t v = *(global_u4_ptr)(unsigned long long)p;
  return make_uint4(v.x, v.y, v.z, v.w);
}
typedef const uint32_t __attribute__((address_space(1)))* global_u32_ptr;
__device__ __forceinline__ uint32_t ld_global4(const void* p) { return *(global_u32_ptr)(unsigned long long)p; }
__device__ __forceinline__ ArcReg load_arc(const wfst_tr* p) {
  const uint4 v = ld_global16(p);  // one 16-B load
  return ArcReg{v.x, v.y, __uint_as_float(v.z), v.w};
}
__device__ __forceinline__ void store_arc(wfst_tr* p, uint32_t il, uint32_t ol, float w, uint32_t ns) {
  *reinterpret_cast<uint4*>(p) = make_uint4(il, ol, __float_as_uint(w), ns);
}
__device__ __forceinline__ ArcReg shfl_arc(const ArcReg& a, uint32_t src) {
  return ArcReg{(uint32_t)__shfl(a.il, src), (uint32_t)__shfl(a.ol, src), __shfl(a.w, src), (uint32_t)__shfl(a.ns, src)};
}

// lower / upper bound on the searched side's key column (superslice lower_bound_by,
// matchers/sorted_matcher.rs:141-142), per lane, for arc blocks larger than one wave.
__device__ inline void equal_range_global(const wfst_tr* arcs, uint32_t n, bool by_ilabel, uint32_t key, uint32_t* lo_out,
                                          uint32_t* cnt_out) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    uint32_t mid = lo + ((hi - lo) >> 1);
    uint32_t k = ld_global4(by_ilabel ? &arcs[mid].ilabel : &arcs[mid].olabel);
    if (k < key) lo = mid + 1; else hi = mid;
  }
  uint32_t first = lo;
  hi = n;
  while (lo < hi) {
    uint32_t mid = lo + ((hi - lo) >> 1);
    uint32_t k = ld_global4(by_ilabel ? &arcs[mid].ilabel : &arcs[mid].olabel);
    if (k <= key) lo = mid + 1; else hi = mid;
  }
  *lo_out = first;
  *cnt_out = lo - first;
}

// Optional per-phase cycle accounting (build with -DWFST_PHASE_TIMING; results land in Result::dbg).
#ifdef WFST_PHASE_TIMING
#define PT_DECL unsigned long long pt_t = clock64(), pt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PT_MARK(k)                              \
  {                                             \
    const unsigned long long pt_n = clock64();  \
    pt_acc[k] += pt_n - pt_t;                   \
    pt_t = pt_n;                                \
  }
#define PT_STORE(res) \
  for (int pt_i = 0; pt_i < 8; ++pt_i) (res).dbg[pt_i] = pt_acc[pt_i];
#else
#define PT_DECL
#define PT_MARK(k)
#define PT_STORE(res)
#endif

struct Level {
  uint32_t arc_begin;  // first emitted arc of the level
  uint32_t hi;         // number of ids assigned before this level's new states
};

// LDS staging of ONE BFS level when it emits <= 64 arcs (the common case for acceptor o transducer
// lattices): the level's arcs, destination tuples and shortest-path candidates stay on-chip, duplicates
// are resolved with lane shuffles, and only first occurrences touch the hash table.
struct FastStage {
  uint32_t il[64], ol[64];
  float w[64];
  uint64_t key[64];   // destination tuple
  uint64_t cand[64];  // shortest-path candidate key of the destination through this arc
  uint64_t nkey[64];  // compacted new states (next frontier): tuple
  uint64_t nsk[64];   //                                        shortest-path key
  uint4 nr1[64];      //                                        state record in fst1 (prefetched)
  uint4 nr2[64];      //                                        state record in fst2 (prefetched)
};

enum : int { EXP_OK = 0, EXP_ARENA_OVERFLOW = 1, EXP_FAST_OVERFLOW = 2 };

// ComposeFstOp::compute_trs + compute_final_weight for composed state q = tuple `tk`
// (compose_fst_op.rs:406-449).
//  FAST = false: appends q's arcs at arcs[*n_arcs ...]; destination = hash slot (patched to the id after
//                the level is ranked).
//  FAST = true : stages the arcs in LDS at stg[*level_cnt ...]; nothing touches the hash table here.
template <bool FAST>
__device__ int expand_state(const FstView& f1, const FstView& f2, uint32_t mode, uint32_t filter, const Arena& ar,
                            const Caps& caps,
                            uint32_t q, uint64_t tk, uint4 r1, uint4 r2, uint64_t src_sk, const Level& lv,
                            uint32_t* n_arcs_io, FastStage* stg, uint32_t* level_cnt_io, uint32_t* status) {
  const uint32_t lane = lane_id();
  const uint32_t hmask = caps.H - 1;
  const uint32_t fs = (uint32_t)(tk >> 62);
  const uint32_t s1 = tuple_s1(tk);
  const uint32_t s2 = (uint32_t)tk;
  // r1 / r2 = the 16-byte state records {arc begin, arc count, final bits, noeps} of s1 in fst1 / s2 in fst2
  const uint32_t b1 = r1.x, b2 = r2.x;
  const uint32_t n1 = r1.y, n2 = r2.y;
  const float fin1 = __uint_as_float(r1.z), fin2 = __uint_as_float(r2.z);
  // compute_final_weight :420-449 (final1 (x) final2; None when either is None / the product is zero)
  if (lane == 0) ar.fin[q] = (fin1 != INF && fin2 != INF) ? wtimes(fin1, fin2) : INF;
  // set_state of the filters (sequence_compose_filter.rs:134-148, alt_sequence_compose_filter.rs:143-158,
  // match_compose_filter.rs:126-147): fst1 looks at its OUTPUT epsilons, fst2 at its INPUT epsilons; the state
  // record carries the four facts as bits (SREC_*)
  const bool alleps1 = (r1.w & SREC_ALL_OEPS) && !(fin1 != INF);
  const bool noeps1 = (r1.w & SREC_NO_OEPS) != 0;
  const bool alleps2 = (r2.w & SREC_ALL_IEPS) && !(fin2 != INF);
  const bool noeps2 = (r2.w & SREC_NO_IEPS) != 0;
  // match_input :199-219 (SortedMatcher::priority = num_trs)
  const bool mi = mode == MODE_BOTH ? (n1 <= n2) : (mode == MODE_INPUT);
  const wfst_tr* it_arcs = mi ? f1.arcs + b1 : f2.arcs + b2;
  const wfst_tr* se_arcs = mi ? f2.arcs + b2 : f1.arcs + b1;
  const uint32_t n_it = mi ? n1 : n2;
  const uint32_t n_se = mi ? n2 : n1;
  const uint32_t sa = mi ? s2 : s1;
  const uint32_t sb = mi ? s1 : s2;
  // filter_tr outcomes, constant per composed state; the four classes of (arc1, arc2) pairs a matcher can produce:
  //   X: arc1.olabel == NO_LABEL (fst1 stands still, fst2 takes an input-epsilon arc)
  //   Y: arc2.ilabel == NO_LABEL (fst2 stands still, fst1 takes an output-epsilon arc)
  //   Z: both real, arc1.olabel == 0 == arc2.ilabel                 M: both real, matching non-epsilon label
  const FilterOutcomes fo = filter_outcomes(filter, fs, alleps1, noeps1, alleps2, noeps2);  // compose_filters.h
  const uint32_t fsX = fo.fsX, fsY = fo.fsY, fsZ = fo.fsZ, fsM = 0u;
  const uint32_t fs_nolabel = mi ? fsX : fsY;  // iterated arc labelled NO_LABEL (the loop pseudo-arc)
  const uint32_t fs_eps = mi ? fsY : fsX;      // iterated arc labelled 0 paired with the matcher's EpsLoop

  const bool small = n_se <= 64;
  ArcReg se{0, 0, 0.0f, 0};
  uint32_t se_key = 0xFFFFFFFEu;
  if (small && lane < n_se) {
    se = load_arc(se_arcs + lane);
    se_key = mi ? se.il : se.ol;
  }
  const float src_d = dec_f32((uint32_t)(src_sk >> 32));
  const uint32_t src_h1 = (uint32_t)src_sk + 1u;

  uint32_t n_arcs = *n_arcs_io;
  uint32_t level_cnt = FAST ? *level_cnt_io : 0u;
  const uint32_t n_items = n_it + 1;  // item 0 = the loop pseudo-arc (ordered_expand :229-233)
  for (uint32_t base = 0; base < n_items; base += 64) {
    const uint32_t j = base + lane;
    const bool have = j < n_items;
    ArcReg ab{0, 0, 0.0f, 0};
    if (have) {
      if (j == 0)
        ab = mi ? ArcReg{0u, NO_LABEL, 0.0f, sb} : ArcReg{NO_LABEL, 0u, 0.0f, sb};
      else
        ab = load_arc(it_arcs + (j - 1));
    }
    const uint32_t label = mi ? ab.ol : ab.il;                    // match_tr :333
    const uint32_t skey = label == NO_LABEL ? 0u : label;          // IteratorSortedMatcher::new :131-135
    uint32_t lo = 0, cnt = 0;
    if (small) {
      const uint32_t in_chunk = min(64u, n_items - base);
      for (uint32_t jj = 0; jj < in_chunk; ++jj) {
        const uint32_t k = rl(skey, jj);
        const uint64_t m = __ballot(lane < n_se && se_key == k);
        if (lane == jj) {
          cnt = (uint32_t)__popcll(m);
          lo = m ? (uint32_t)__ffsll((unsigned long long)m) - 1u : 0u;
        }
      }
    } else if (have) {
      equal_range_global(se_arcs, n_se, mi, skey, &lo, &cnt);
    }
    uint32_t fsn;  // filter state of the emitted arcs of this item (of its real pairs when eps_item)
    bool eps_item = false;
    uint32_t loop1 = 0;  // 1: the first pair of an eps_item is the matcher's EpsLoop (accepted by the filter)
    if (!have) {
      cnt = 0;
      fsn = REJECT;
    } else if (label == NO_LABEL) {
      fsn = fs_nolabel;  // real arcs with key 0, no EpsLoop (sorted_matcher.rs:127-139)
    } else if (label == 0u) {
      // the matcher yields EpsLoop first, then the real epsilon arcs (:124-184); the filter sees the loop as class
      // X/Y and the real ones as class Z (rejected by the sequence filters, accepted by Null/Trivial/Match)
      eps_item = true;
      loop1 = fs_eps != REJECT ? 1u : 0u;
      fsn = fsZ;
      cnt = loop1 + (fsZ != REJECT ? cnt : 0u);
    } else {
      fsn = fsM;
    }
    if (!eps_item && fsn == REJECT) cnt = 0;
    uint32_t total, pos, maxcnt;
    if (__ballot(cnt > 1u) == 0) {  // every item emits 0 or 1 arc (the usual case): two ballots replace the scans
      const uint64_t em = __ballot(cnt == 1u);
      pos = lanes_below(em);
      total = (uint32_t)__popcll(em);
      maxcnt = total ? 1u : 0u;
    } else {
      pos = wave_excl_scan(cnt, lane, &total);
      maxcnt = wave_max(cnt);
    }
    if (total == 0) continue;
    if (FAST && level_cnt + total > 64u) return EXP_FAST_OVERFLOW;
    if ((FAST ? n_arcs + level_cnt : n_arcs) + total > caps.A) {
      *status = ST_OVERFLOW_ARCS;
      return EXP_ARENA_OVERFLOW;
    }
    if ((uint64_t)lv.hi + (FAST ? level_cnt : n_arcs - lv.arc_begin) + total + 64 > (uint64_t)caps.H) {
      *status = ST_OVERFLOW_HASH;
      return EXP_ARENA_OVERFLOW;
    }
    for (uint32_t m = 0; m < maxcnt; ++m) {
      ArcReg aa;
      const bool is_loop = loop1 != 0u && m == 0u;  // (loop1 is only ever set on eps items)
      if (small) {
        aa = shfl_arc(se, (lo + m - loop1) & 63u);
      } else {
        aa = ArcReg{0, 0, 0.0f, 0};
        if (m < cnt && !is_loop) aa = load_arc(se_arcs + lo + m - loop1);
      }
      if (m < cnt) {
        const uint32_t fsn_m = is_loop ? fs_eps : fsn;
        if (is_loop) aa = mi ? ArcReg{NO_LABEL, 0u, 0.0f, sa} : ArcReg{0u, NO_LABEL, 0.0f, sa};  // eps_loop, mod.rs:98-105
        // arc1 from fst1, arc2 from fst2 (match_tr_selected :301-319); selected field by field so that the
        // structs stay in registers (a reference select would push them to scratch)
        const ArcReg a1{mi ? ab.il : aa.il, mi ? ab.ol : aa.ol, mi ? ab.w : aa.w, mi ? ab.ns : aa.ns};
        const ArcReg a2{mi ? aa.il : ab.il, mi ? aa.ol : ab.ol, mi ? aa.w : ab.w, mi ? aa.ns : ab.ns};
        const uint64_t key = pack_tuple(fsn_m, a1.ns, a2.ns);
        const float wsum = wtimes(a1.w, a2.w);  // add_tr :267-285
        if (FAST) {
          const uint32_t e = level_cnt + pos + m;
          stg->il[e] = a1.il;
          stg->ol[e] = a2.ol;
          stg->w[e] = wsum;
          stg->key[e] = key;
          uint64_t cand = KEY_INF;
          if (src_sk != KEY_INF) {
            const float c = (src_d + wsum) + 0.0f;
            if (c < INF) cand = ((uint64_t)enc_f32(c) << 32) | src_h1;
          }
          stg->cand[e] = cand;
        } else {
          const uint32_t e = n_arcs + pos + m;
          const uint32_t slot = ht_insert(ar.hkeys, ar.hvals, hmask, key, HV_PEND | (e - lv.arc_begin));
          store_arc(ar.arcs + e, a1.il, a2.ol, wsum, slot);
        }
      }
    }
    if (FAST)
      level_cnt += total;
    else
      n_arcs += total;
  }
  if (FAST)
    *level_cnt_io = level_cnt;
  else
    *n_arcs_io = n_arcs;
  return EXP_OK;
}

// StateTable::find_id for a tuple seen for the first time in this level (fast path): CAS first, so a
// new tuple costs one L2 round trip.  Returns the slot; *existed tells whether an earlier level owns it.
__device__ __forceinline__ uint32_t ht_find_or_insert(uint64_t* hkeys, uint32_t hmask, uint64_t key, bool* existed) {
  uint32_t slot = hash_u64(key) & hmask;
  for (uint32_t probes = 0; probes <= hmask; ++probes) {
    const uint64_t prev = atomicCAS((unsigned long long*)&hkeys[slot], (unsigned long long)HT_EMPTY,
                                    (unsigned long long)key);
    if (prev == HT_EMPTY) {
      *existed = false;
      return slot;
    }
    if (prev == key) {
      *existed = true;
      return slot;
    }
    slot = (slot + 1) & hmask;
  }
  *existed = false;
  return slot;
}

// One relaxation pass over states [lo, hi): lanes over states.  Returns (wave-uniform) whether any key improved.
__device__ bool relax_states(const Arena& ar, uint32_t lo, uint32_t hi, uint32_t n_arcs_total) {
  const uint32_t lane = lane_id();
  bool changed = false;
  for (uint32_t base = lo; base < hi; base += 64) {
    const uint32_t q = base + lane;
    if (q < hi) {
      const uint64_t kq = ld_l2(&ar.skey[q]);
      if (kq != KEY_INF) {
        const float d = dec_f32((uint32_t)(kq >> 32));
        const uint32_t h1 = (uint32_t)kq + 1u;
        const uint32_t b = ar.off[q], e = ar.off[q + 1];
        for (uint32_t i = b; i < e && i < n_arcs_total; ++i) {
          const ArcReg a = load_arc(ar.arcs + i);
          const float c = (d + a.w) + 0.0f;
          if (!(c < INF)) continue;
          const uint64_t ck = ((uint64_t)enc_f32(c) << 32) | h1;
          const uint64_t old = atomicMin((unsigned long long*)&ar.skey[a.ns], (unsigned long long)ck);
          if (ck < old) changed = true;
        }
      }
    }
  }
  return __any(changed);
}

template <uint32_t FLAGS>
__global__ void __launch_bounds__(64) compose_wave_kernel(const ProblemDesc* __restrict__ descs, FstView f2, Caps caps,
                                                          char* __restrict__ arena_base, size_t arena_stride,
                                                          Result* __restrict__ results, wfst_tr* __restrict__ path_buf,
                                                          uint32_t path_cap, uint32_t* __restrict__ path_cursor) {
  const uint32_t p = blockIdx.x;
  const uint32_t lane = lane_id();
  const ProblemDesc desc = descs[p];
  const FstView f1 = desc.f1;
  const uint32_t mode = desc.mode;
  const uint32_t filter = desc.filter;
  const Arena ar = carve_arena(arena_base + (size_t)p * arena_stride, caps);
  Result res;
  res.status = ST_OK;
  res.n_states = res.n_arcs = res.t_states = res.t_arcs = 0;
  res.t_start = -1;
  res.has_path = res.hops = 0;
  res.final_weight = res.total = INF;
  res.path_off = 0;
  res.n_levels = 0;
  res.facts = props::PATH_FACTS_NONE;

  __shared__ FastStage stg;
  PT_DECL
  uint32_t n_states = 0, n_arcs = 0, n_levels = 0;
  bool ok = true;
  bool needs_fixup = false;

  // compute_start :389-404
  if (f1.start >= 0 && f2.start >= 0) {
    // clear this problem's table
    for (uint32_t i = lane; i < caps.H; i += 64) {
      ar.hkeys[i] = HT_EMPTY;
      ar.hvals[i] = HV_INIT;
    }
    if (FLAGS & FLAG_SP)
      for (uint32_t i = lane; i < caps.S; i += 64) ar.skey[i] = KEY_INF;
    wave_sync();
    if (lane == 0) {
      const uint64_t key0 = pack_tuple(0u, (uint32_t)f1.start, (uint32_t)f2.start);
      const uint32_t slot = ht_insert(ar.hkeys, ar.hvals, caps.H - 1, key0, 0u);
      st_l2(&ar.hvals[slot], 0u);
      ar.tuples[0] = key0;
      if (FLAGS & FLAG_SP) ar.skey[0] = (uint64_t)enc_f32(0.0f) << 32;
    }
    wave_sync();
    n_states = 1;
    uint32_t lo = 0, hi = 1;
    PT_MARK(0)  // table clear + start tuple
    // frontier in registers while it is <= 64 states wide: lane i holds state lo + i
    bool fast = true;
    uint64_t f_key = lane == 0 ? pack_tuple(0u, (uint32_t)f1.start, (uint32_t)f2.start) : 0ull;
    uint64_t f_sk = lane == 0 ? ((uint64_t)enc_f32(0.0f) << 32) : KEY_INF;
    uint4 f_r1 = make_uint4(0, 0, 0, 0), f_r2 = make_uint4(0, 0, 0, 0);
    if (lane == 0) {
      f_r1 = ld_global16(f1.srec + f1.start);
      f_r2 = ld_global16(f2.srec + f2.start);
    }
    // LazyFst::compute :235-259 — FIFO BFS; level k = ids [lo, hi)
    while (lo < hi && ok) {
      if (lane == 0) ar.lvl[n_levels] = lo;
      n_levels++;
      Level lv{n_arcs, hi};
      bool did_fast = false;
      if (fast) {
        // ---------------- small-level fast path: the whole level lives in LDS / registers
        uint32_t level_cnt = 0;
        int rc = EXP_OK;
        for (uint32_t q = lo; q < hi; ++q) {
          const uint64_t tk = rl64(f_key, q - lo);
          const uint64_t ssk = rl64(f_sk, q - lo);
          if (lane == 0) ar.off[q] = n_arcs + level_cnt;
          rc = expand_state<true>(f1, f2, mode, filter, ar, caps, q, tk, rl128(f_r1, q - lo), rl128(f_r2, q - lo), ssk, lv,
                                  &n_arcs, &stg, &level_cnt, &res.status);
          if (rc != EXP_OK) break;
        }
        if (rc == EXP_ARENA_OVERFLOW) {
          ok = false;
          break;
        }
        PT_MARK(1)  // fast: expansions (state records -> arc blocks -> matches -> LDS stage)
        if (rc == EXP_OK) {
          did_fast = true;
          lds_handoff();
          const bool have = lane < level_cnt;
          uint32_t il = 0, ol = 0;
          float w = 0.0f;
          uint64_t key = HT_EMPTY, cand = KEY_INF;
          if (have) {
            il = stg.il[lane];
            ol = stg.ol[lane];
            w = stg.w[lane];
            key = stg.key[lane];
            cand = stg.cand[lane];
          }
          // speculative fetch of the destination's state records: overlaps the hash-table round trip below and
          // removes one dependent trip to HBM from the next level
          uint4 pr1 = make_uint4(0, 0, 0, 0), pr2 = make_uint4(0, 0, 0, 0);
          if (have) {
            pr1 = ld_global16(f1.srec + tuple_s1(key));
            pr2 = ld_global16(f2.srec + (uint32_t)key);
          }
          // first occurrence of each destination tuple inside the level + min candidate of its group
          uint64_t mymin = cand;
          uint32_t first = 64;
          for (uint32_t i = 0; i < level_cnt; ++i) {
            const uint64_t ki = rl64(key, i);
            const uint64_t ci = rl64(cand, i);
            if (have && ki == key) {
              mymin = ci < mymin ? ci : mymin;
              if (first == 64) first = i;
            }
          }
          const bool is_first = have && first == lane;
          PT_MARK(2)  // fast: stage read + record prefetch issue + in-register dedupe
          uint32_t slot = 0, id = 0;
          bool existed = false;
          if (is_first) {
            slot = ht_find_or_insert(ar.hkeys, caps.H - 1, key, &existed);
            if (existed) id = ld_l2(&ar.hvals[slot]);
          }
          const bool newf = is_first && !existed;
          const uint64_t nm = __ballot(newf);
          const uint32_t n_new = (uint32_t)__popcll(nm);
          if (hi + n_new > caps.S) {
            res.status = ST_OVERFLOW_STATES;
            ok = false;
            break;
          }
          if (newf) {  // new id = hi + (number of earlier first occurrences)  [state_table.rs:49-59]
            const uint32_t rank = lanes_below(nm);
            id = hi + rank;
            ar.hvals[slot] = id;  // plain store: later readers are this wave (same CU / L2); an sc1 store's ack is slow
            ar.tuples[id] = key;
            if (FLAGS & FLAG_SP) ar.skey[id] = mymin;
            stg.nkey[rank] = key;
            stg.nsk[rank] = mymin;
            stg.nr1[rank] = pr1;
            stg.nr2[rank] = pr2;
          } else if ((FLAGS & FLAG_SP) && is_first && mymin != KEY_INF) {
            atomicMin((unsigned long long*)&ar.skey[id], (unsigned long long)mymin);  // older state: fix-up will iterate
          }
          PT_MARK(3)  // fast: hash CAS + id assignment + stores issue
          const uint32_t id_all = __shfl(id, first & 63u);
          if (have) {
            store_arc(ar.arcs + n_arcs + lane, il, ol, w, id_all);
            if (id_all < hi) needs_fixup = true;  // arc into this or an earlier level
          }
          n_arcs += level_cnt;
          if (lane == 0) ar.off[hi] = n_arcs;
          lds_handoff();
          if (lane < n_new) {
            f_key = stg.nkey[lane];
            f_sk = stg.nsk[lane];
            f_r1 = stg.nr1[lane];
            f_r2 = stg.nr2[lane];
          } else {
            f_key = 0ull;
            f_sk = KEY_INF;
          }
          lds_handoff();
          lo = hi;
          hi += n_new;
          n_states = hi;
          PT_MARK(4)  // fast: arc stores + frontier compaction
        }
        // rc == EXP_FAST_OVERFLOW: more than 64 arcs in this level -> redo it on the general path
      }
      if (did_fast) continue;
      // ---------------- general path: level staged through the arena in HBM/L2
      for (uint32_t q = lo; q < hi; ++q) {
        if (lane == 0) ar.off[q] = n_arcs;
        const uint64_t tk = ar.tuples[q];
        if (expand_state<false>(f1, f2, mode, filter, ar, caps, q, tk, ld_global16(f1.srec + tuple_s1(tk)),
                                ld_global16(f2.srec + (uint32_t)tk), KEY_INF, lv, &n_arcs, nullptr, nullptr, &res.status) != EXP_OK) {
          ok = false;
          break;
        }
      }
      if (!ok) break;
      if (lane == 0) ar.off[hi] = n_arcs;
      wave_sync();
      // rank first occurrences: new id = hi + (number of earlier first occurrences)  [state_table.rs:49-59]
      uint32_t n_new = 0;
      for (uint32_t base = lv.arc_begin; base < n_arcs; base += 64) {
        const uint32_t e = base + lane;
        bool first = false;
        uint32_t slot = 0;
        if (e < n_arcs) {
          slot = ar.arcs[e].nextstate;
          first = ld_l2(&ar.hvals[slot]) == (HV_PEND | (e - lv.arc_begin));
        }
        const uint64_t m = __ballot(first);
        const uint32_t cntm = (uint32_t)__popcll(m);
        if (hi + n_new + cntm > caps.S) {
          res.status = ST_OVERFLOW_STATES;
          ok = false;
          break;
        }
        if (first) {
          const uint32_t id = hi + n_new + lanes_below(m);
          st_l2(&ar.hvals[slot], id);
          ar.tuples[id] = ld_l2(&ar.hkeys[slot]);
        }
        n_new += cntm;
      }
      if (!ok) break;
      wave_sync();
      // patch destinations: slot -> id
      for (uint32_t base = lv.arc_begin; base < n_arcs; base += 64) {
        const uint32_t e = base + lane;
        if (e < n_arcs) {
          const uint32_t id = ld_l2(&ar.hvals[ar.arcs[e].nextstate]);
          ar.arcs[e].nextstate = id;
          if (id < hi) needs_fixup = true;  // arc into this or an earlier level: one forward pass is not enough
        }
      }
      wave_sync();
      if (FLAGS & FLAG_SP) {
        // relax this level's arcs now: on a lattice every predecessor of these states is already final
        relax_states(ar, lo, hi, n_arcs);
        wave_sync();
      }
      if (caps.W && n_new > caps.W) {
        res.status = ST_SWITCH_WIDE;
        ok = false;
        break;
      }
      lo = hi;
      hi += n_new;
      n_states = hi;
      fast = n_new <= 64;
      if (fast) {  // pull the new frontier back into registers
        f_key = lane < n_new ? ar.tuples[lo + lane] : 0ull;
        f_sk = ((FLAGS & FLAG_SP) && lane < n_new) ? ld_l2(&ar.skey[lo + lane]) : KEY_INF;
        if (lane < n_new) {
          f_r1 = ld_global16(f1.srec + tuple_s1(f_key));
          f_r2 = ld_global16(f2.srec + (uint32_t)f_key);
        }
      }
    }
    if (lane == 0) ar.lvl[n_levels] = n_states;
    PT_MARK(5)  // general-path levels
  }
  needs_fixup = __any(needs_fixup);
  res.n_states = n_states;
  res.n_arcs = n_arcs;
  res.n_levels = n_levels;
  res.t_states = n_states;
  res.t_arcs = n_arcs;
  res.t_start = n_states ? 0 : -1;

  // ------------------------------------------------------------ trim: connect() + del_states (stable)
  if ((FLAGS & FLAG_TRIM) && ok && n_states) {
    // every BFS-discovered state is accessible; coaccess = can reach a final state (connect.rs:54-60)
    for (uint32_t i = lane; i < n_states; i += 64) ar.aux[i] = ar.fin[i] != INF ? 1u : 0u;
    wave_sync();
    bool changed = true;
    while (changed) {
      changed = false;
      for (int32_t k = (int32_t)n_levels - 1; k >= 0; --k) {  // reverse level order: one pass on lattices
        const uint32_t llo = ar.lvl[k], lhi = ar.lvl[k + 1];
        for (uint32_t base = llo; base < lhi; base += 64) {
          const uint32_t q = base + lane;
          if (q < lhi && ld_l2(&ar.aux[q]) == 0u) {
            const uint32_t b = ar.off[q], e = ar.off[q + 1];
            for (uint32_t i = b; i < e; ++i) {
              if (ld_l2(&ar.aux[ar.arcs[i].nextstate])) {
                st_l2(&ar.aux[q], 1u);
                changed = true;
                break;
              }
            }
          }
        }
        wave_sync();
      }
      changed = __any(changed);
    }
    // new ids = rank among survivors (mutable_fst.rs:139-149)
    uint32_t kept = 0;
    for (uint32_t base = 0; base < n_states; base += 64) {
      const uint32_t q = base + lane;
      const bool keep = q < n_states && ld_l2(&ar.aux[q]) != 0u;
      const uint64_t m = __ballot(keep);
      if (q < n_states) ar.aux2[q] = keep ? kept + lanes_below(m) : REJECT;
      kept += (uint32_t)__popcll(m);
    }
    wave_sync();
    // surviving arcs, order preserved, retargeted (mutable_fst.rs:153-176)
    uint32_t t_arcs = 0;
    for (uint32_t base = 0; base < n_states; base += 64) {
      const uint32_t q = base + lane;
      const bool keep = q < n_states && ar.aux2[q] != REJECT;
      uint32_t b = 0, e = 0, cnt = 0;
      if (keep) {
        b = ar.off[q];
        e = ar.off[q + 1];
        for (uint32_t i = b; i < e; ++i) cnt += ar.aux2[ar.arcs[i].nextstate] != REJECT;
      }
      uint32_t total;
      const uint32_t pos = wave_excl_scan(cnt, lane, &total);
      if (keep) {
        const uint32_t nq = ar.aux2[q];
        uint32_t w = t_arcs + pos;
        ar.t_off[nq] = w;
        ar.t_fin[nq] = ar.fin[q];
        for (uint32_t i = b; i < e; ++i) {
          const ArcReg a = load_arc(ar.arcs + i);
          const uint32_t nn = ar.aux2[a.ns];
          if (nn != REJECT) store_arc(ar.t_arcs + (w++), a.il, a.ol, a.w, nn);
        }
      }
      t_arcs += total;
    }
    if (lane == 0) ar.t_off[kept] = t_arcs;
    res.t_states = kept;
    res.t_arcs = t_arcs;
    res.t_start = (kept && ar.aux2[0] != REJECT) ? (int32_t)ar.aux2[0] : -1;
    wave_sync();
  }

  // ------------------------------------------------------------ fused n=1 shortest path on the composed FST
  // (run on the untrimmed composition: dead states cannot lie on a successful path, and the canonical tie
  //  rule is invariant under the stable renumbering of trim, so the result equals shortest_path(connect(C)))
  if ((FLAGS & FLAG_SP) && ok && n_states) {
    if (needs_fixup) {  // cyclic / non-layered composition: iterate to the fixed point
      for (uint32_t it = 0; it <= n_states; ++it) {
        wave_sync();
        if (!relax_states(ar, 0, n_states, n_arcs)) break;
      }
    }
    wave_sync();
    // f_parent = argmin (d (x) rho, id)      shortest_path.rs:214-220
    unsigned long long best = KEY_INF;
    for (uint32_t base = 0; base < n_states; base += 64) {
      const uint32_t q = base + lane;
      if (q < n_states) {
        const uint64_t kq = ld_l2(&ar.skey[q]);
        const float f = ar.fin[q];
        if (kq != KEY_INF && f < INF) {
          const float tot = (dec_f32((uint32_t)(kq >> 32)) + f) + 0.0f;
          if (tot < INF) {
            const unsigned long long c = ((unsigned long long)enc_f32(tot) << 32) | q;
            best = c < best ? c : best;
          }
        }
      }
    }
    best = wave_min_u64(best);
    if (best != KEY_INF) {
      const uint32_t fp = (uint32_t)best;
      const uint32_t hops = (uint32_t)ld_l2(&ar.skey[fp]);
      res.has_path = 1;
      res.hops = hops;
      res.final_weight = ar.fin[fp];
      res.total = dec_f32((uint32_t)(best >> 32));
      // parent[t] = min (s,pos) among layered tight arcs
      for (uint32_t i = lane; i < n_states; i += 64) ar.parent[i] = KEY_INF;
      wave_sync();
      for (uint32_t base = 0; base < n_states; base += 64) {
        const uint32_t q = base + lane;
        if (q < n_states) {
          const uint64_t kq = ld_l2(&ar.skey[q]);
          if (kq != KEY_INF) {
            const float d = dec_f32((uint32_t)(kq >> 32));
            const uint32_t h1 = (uint32_t)kq + 1u;
            const uint32_t b = ar.off[q], e = ar.off[q + 1];
            for (uint32_t i = b; i < e; ++i) {
              const ArcReg a = load_arc(ar.arcs + i);
              const float c = (d + a.w) + 0.0f;
              if (!(c < INF)) continue;
              const uint64_t ck = ((uint64_t)enc_f32(c) << 32) | h1, kt = ld_l2(&ar.skey[a.ns]);
              // class 0: tight in distance and hop count; class 1 (top bit): tight in the distance only, from a state
              // with a smaller key (sssp.hip parent_class)
              if (ck == kt)
                atomicMin((unsigned long long*)&ar.parent[a.ns], ((unsigned long long)q << 32) | (i - b));
              else if ((ck >> 32) == (kt >> 32) && kq < kt)
                atomicMin((unsigned long long*)&ar.parent[a.ns], (1ull << 63) | ((unsigned long long)q << 32) | (i - b));
            }
          }
        }
      }
      wave_sync();
      // mark the ancestors of f_parent by pointer doubling: aux = mark, jump pointers ping-pong
      // between aux2 and lvl (dead after trim)
      uint32_t* ja = ar.aux2;
      uint32_t* jb = ar.lvl;
      for (uint32_t i = lane; i < n_states; i += 64) {
        const uint64_t pr = ld_l2(&ar.parent[i]);
        ja[i] = pr == KEY_INF ? i : ((uint32_t)(pr >> 32) & 0x7FFFFFFFu);
        ar.aux[i] = i == fp ? 1u : 0u;
      }
      wave_sync();
      for (uint32_t span = 1; span <= hops; span <<= 1) {
        // mark[jump[t]] |= mark[t]; jump'[t] = jump[jump[t]]
        for (uint32_t i = lane; i < n_states; i += 64) {
          const uint32_t j = ja[i];
          if (ld_l2(&ar.aux[i])) st_l2(&ar.aux[j], 1u);
          jb[i] = ja[j];
        }
        wave_sync();
        uint32_t* tmp = ja;
        ja = jb;
        jb = tmp;
      }
      // reserve the slice of the packed path buffer and write the arcs (shortest_path.rs:257-272)
      uint32_t poff = 0;
      if (lane == 0) poff = atomicAdd(path_cursor, hops);
      poff = __shfl(poff, 0);
      // a marked state without a class-0 predecessor: positions cannot be read off the hop counts
      bool tie = false;
      for (uint32_t i = lane; i < n_states; i += 64)
        if (ld_l2(&ar.aux[i]) && (uint32_t)ld_l2(&ar.skey[i]) >= 1u && (ld_l2(&ar.parent[i]) >> 63)) tie = true;
      if (__any(tie)) {
        res.status = ST_TIE_ORDER;
      } else if ((uint64_t)poff + hops > path_cap) {
        res.status = ST_OVERFLOW_PATH;
      } else {
        res.path_off = poff;
        for (uint32_t i = lane; i < n_states; i += 64) {
          if (ld_l2(&ar.aux[i])) {
            const uint64_t ki = ld_l2(&ar.skey[i]);
            const uint32_t hi_ = (uint32_t)ki;
            if (hi_ >= 1u) {
              const uint64_t pr = ld_l2(&ar.parent[i]);
              const uint32_t s = (uint32_t)(pr >> 32), pos = (uint32_t)pr;
              const ArcReg a = load_arc(ar.arcs + ar.off[s] + pos);
              const uint32_t k = hops - hi_;  // i is the k-th state created by the backtrace
              store_arc(path_buf + poff + k, a.il, a.ol, a.w, k);
            }
          }
        }
      }
    }
  }
  PT_MARK(6)  // trim / shortest-path epilogue
  PT_STORE(res)
  if (lane == 0) results[p] = res;
}


// ------------------------------------------------------------------------------------------------------------------
// string o T  ->  shortest path, one wave per problem: the decoding case of the fused batch, where fst1 is a linear,
// epsilon-free, single-final acceptor (utils::acceptor, labels_to_fst.rs:111-132) and fst2 has no input-epsilon arcs.
// The composition is then a layered lattice — BFS level i = the composed states (i, q) — so the general machinery above
// collapses:
//   * a tuple first seen in level i + 1 cannot have been seen before: first-occurrence ranking inside the level (ballots over
//     the new states, which live in lane registers) IS StateTable::find_id; no hash table, no CAS round trip;
//   * ids are handed out level by level in emission order, arcs of a state in fst2's arc order (the label-equal run of a
//     sorted block): the numbering LazyFst::compute produces (lazy/lazy_fst.rs:226-269), hence the same canonical parent
//     (smallest (source id, arc position) among the tight arcs == the first strict improvement in emission order);
//   * every arc goes from level i to level i + 1: one forward pass is the exact (min,+) fixed point, hops == level;
//   * the arc block of a destination is named by the matched arc itself (FstView::anext), so a level costs ONE dependent
//     trip to memory (the arc blocks of the frontier) instead of three (state record -> arc block -> hash).
// Composed states / arcs are only counted (the fused batch returns paths); parents live in LDS.  Anything outside the
// covered case (more than 64 states in a level, more than STR_MAXS states in all) reports ST_NOT_A_STRING_CASE and the
// host re-runs that problem on compose_wave_kernel.  Results are bit-identical to the general kernel's by construction and
// by test (tests/test_gpu_parity.py: both kernels against the oracle).
constexpr uint64_t WIDE_COMPOSE_STATES = 16384;  // compose(): results beyond this go to compose_wide.hip
constexpr uint32_t WIDE_COMPOSE_WIDTH = 64;      // ... and so do results with a BFS level wider than this (one wave's worth)
constexpr uint32_t STR_MAXS = 2048;
constexpr uint32_t STR_NONE = 0xFFFFFFFFu;

// One wave = one problem; a workgroup holds blockDim.x / 64 of them, each with its own `maxs`-state slice of the dynamic
// LDS (the waves never meet: no workgroup barrier anywhere).  Why several waves per workgroup: a resident wave of this
// kernel blocks its whole CU for the 1024-thread, 128-VGPR workgroups of the mailbox relaxation sweeps (they need every
// register of all four SIMDs), so 64 lone waves on 64 CUs sent a quarter of every concurrent sweep into a second round;
// 8 workgroups of 8 waves land on one CU per XCD and leave 31 CUs per XCD to the sweeps' 30-31 workgroups per XCD.
__global__ void __launch_bounds__(512) string_compose_sp_kernel(const ProblemDesc* __restrict__ descs, FstView f2,
                                                                Result* __restrict__ results, wfst_tr* __restrict__ path_buf,
                                                                uint32_t path_cap, uint32_t* __restrict__ path_cursor,
                                                                uint32_t n_problems, uint32_t maxs, uint64_t f2_n_arcs,
                                                                uint32_t scalar_rows, uint32_t done_ticket) {
  extern __shared__ uint32_t s_dyn[];
  const uint32_t p = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (p >= n_problems) return;
  uint32_t* const s_par = s_dyn + (size_t)(threadIdx.x >> 6) * 4u * maxs;  // predecessor state on the best path into this state (STR_NONE: unreached)
  uint32_t* const s_ol = s_par + maxs;                                      // olabel of the arc taken from it
  float* const s_w = reinterpret_cast<float*>(s_ol + maxs);                 // weight of that arc (w1 (x) w2)
  uint32_t* const s_pth = s_ol + 2u * maxs;                                 // states of the best path, from the final one backwards
  const uint32_t lane = lane_id();
  // the problem's result, then (everything this wave wrote is in host memory) its ticket
#define STR_RESULT() do {                                                                                  \
    res.done = 0u;                                                                                         \
    if (lane == 0) results[p] = res;                                                                       \
    if (done_ticket != 0u) { /* (0: a large batch — the host waits for the stream, see launch_begin) */     \
      __threadfence_system(); /* (the stores' acknowledgements alone are not enough: measured) */             \
      if (lane == 0) __hip_atomic_store(&results[p].done, done_ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); \
    }                                                                                                      \
  } while (0)
  const FstView f1 = descs[p].f1;
  Result res;
  res.status = ST_OK;
  res.n_states = res.n_arcs = res.t_states = res.t_arcs = 0;
  res.t_start = -1;
  res.has_path = res.hops = 0;
  res.final_weight = res.total = INF;
  res.path_off = 0;
  res.n_levels = 0;
  res.facts = props::PATH_FACTS_NONE;
  if (f1.start < 0 || f2.start < 0) {  // compute_start -> None
    STR_RESULT();
    return;
  }
  const uint32_t L = f1.n_states - 1u;  // number of arcs of the string
  // frontier of the current level in registers: lane k < F holds composed state lo + k = (pos, q)
  uint32_t q = 0, nb = 0, nc = 0;
  float d = INF;
  if (lane == 0) {
    q = (uint32_t)f2.start;
    d = 0.0f;  // Weight::one()
    const uint4 r = ld_global16(f2.srec + q);
    nb = r.x;
    nc = r.y;
  }
  uint32_t lo = 0, hi = 1, F = 1, n_arcs = 0, level = 0;  // level = position of the current frontier in the string
  if (lane == 0) s_par[0] = STR_NONE;
  // the string's arcs, 64 at a time, one per lane (a linear acceptor: arc i leaves state i), the next 64 already on their
  // way: a label is a v_readlane away.  (One broadcast load per level, asked for one level ahead, made every level wait
  // for a vector load — ~0.6 us for a lone wave — however fast the arc block of fst2 came in.)
  uint4 ch_cur = make_uint4(0, 0, 0, 0), ch_nxt = make_uint4(0, 0, 0, 0);
  if (lane < L) ch_cur = ld_global16(f1.arcs + lane);
  if (64u + lane < L) ch_nxt = ld_global16(f1.arcs + 64u + lane);
  bool ok = true;
  for (uint32_t pos = 0; pos < L && F && ok; ++pos) {
    const uint32_t pl = pos & 63u;
    if (pos && pl == 0u) {
      ch_cur = ch_nxt;
      ch_nxt = make_uint4(0, 0, 0, 0);
      if (pos + 64u + lane < L) ch_nxt = ld_global16(f1.arcs + pos + 64u + lane);
    }
    const uint32_t label = rl(ch_cur.x, pl);
    const float w1 = __uint_as_float(rl(ch_cur.z, pl));
    uint32_t nq = 0, nnb = 0, nnc = 0, n_new = 0;  // new states of the next level: lane k holds state hi + k
    float nd = INF;
    // one matched arc (f -> qd, output label ol, weight w2, the destination's arc block [xb, xb + xc)): the destination
    // becomes a state of the next level unless the level has it already, and keeps the better way in
    auto on_match = [&](uint32_t f, float fd, uint32_t qd, uint32_t ol, uint32_t xb, uint32_t xc, float w2) -> bool {
      const float wsum = wtimes(w1, w2);  // add_tr, compose_fst_op.rs:267-285
      float c = INF;
      if (fd < INF) c = (fd + wsum) + 0.0f;  // candidate distance; +inf never relaxes (shortest_path.rs:226)
      const uint64_t ex = __ballot(lane < n_new && nq == qd);  // StateTable::find_id inside the level
      uint32_t idx;
      if (ex == 0) {
        idx = n_new;
        if (idx >= 64u || hi + idx >= maxs) return false;
        if (lane == idx) {
          nq = qd;
          nnb = xb;
          nnc = xc;
          nd = INF;
        }
        if (lane == 0) s_par[hi + idx] = STR_NONE;
        n_new += 1;
      } else {
        idx = (uint32_t)__ffsll((unsigned long long)ex) - 1u;
      }
      const float cur = __uint_as_float(rl(__float_as_uint(nd), idx));
      if (c < cur) {  // strict: on ties the earlier (source id, arc position) stays
        if (lane == idx) nd = c;
        if (lane == 0) {
          s_par[hi + idx] = lo + f;
          s_ol[hi + idx] = ol;
          s_w[hi + idx] = wsum;
        }
      }
      return true;
    };
    // The common level: ONE frontier state with a short arc block.  A lone wave's vector load takes ~0.6 us from issue to
    // use on this chip, a scalar load of the same (immutable) rows ~0.1 us (tools/ubench_chase.hip): the labels of a block
    // of up to 12 arcs come through the scalar cache (the loads also bring every line of the block and of its
    // destination ranges in), they are compared on the scalar unit, the matched arc is a scalar-cache hit, and the next
    // level's block address never leaves SGPRs.
    bool took_scalar = false;
    if (scalar_rows && F == 1u) {
      const uint32_t fb = rl(nb, 0), fc = rl(nc, 0);
      if (fc <= 12u && (uint64_t)fb + 12u <= f2_n_arcs) {
        took_scalar = true;
        const float fd = __uint_as_float(rl(__float_as_uint(d), 0));
        // step 1: the twelve labels (one dword per arc: every 64-byte line of the block is touched) and one word of every line
        // of the destination-range block, all issued together: one miss latency
        uint32_t l0, l1, l2, l3, l4, l5, l6, l7, l8, l9, l10, l11, t0, t1, t2;
        const wfst_tr* ap = f2.arcs + fb;
        const uint2* np = f2.anext + fb;
        asm volatile(
            "s_load_dword %0, %15, 0x0\n\ts_load_dword %1, %15, 0x10\n\ts_load_dword %2, %15, 0x20\n\ts_load_dword %3, %15, 0x30\n\t"
            "s_load_dword %4, %15, 0x40\n\ts_load_dword %5, %15, 0x50\n\ts_load_dword %6, %15, 0x60\n\ts_load_dword %7, %15, 0x70\n\t"
            "s_load_dword %8, %15, 0x80\n\ts_load_dword %9, %15, 0x90\n\ts_load_dword %10, %15, 0xa0\n\ts_load_dword %11, %15, 0xb0\n\t"
            "s_load_dword %12, %16, 0x0\n\ts_load_dword %13, %16, 0x2c\n\ts_load_dword %14, %16, 0x58\n\ts_waitcnt lgkmcnt(0)"
            : "=&s"(l0), "=&s"(l1), "=&s"(l2), "=&s"(l3), "=&s"(l4), "=&s"(l5), "=&s"(l6), "=&s"(l7), "=&s"(l8), "=&s"(l9), "=&s"(l10),
              "=&s"(l11), "=&s"(t0), "=&s"(t1), "=&s"(t2)
            : "s"(ap), "s"(np)
            : "memory");
        uint32_t m = 0;  // arcs of the block that carry the label, in arc order
        m |= (l0 == label ? 1u : 0u) | (l1 == label ? 2u : 0u) | (l2 == label ? 4u : 0u) | (l3 == label ? 8u : 0u);
        m |= (l4 == label ? 16u : 0u) | (l5 == label ? 32u : 0u) | (l6 == label ? 64u : 0u) | (l7 == label ? 128u : 0u);
        m |= (l8 == label ? 256u : 0u) | (l9 == label ? 512u : 0u) | (l10 == label ? 1024u : 0u) | (l11 == label ? 2048u : 0u);
        m &= (1u << fc) - 1u;
        n_arcs += (uint32_t)__popc(m);
        while (m) {
          const uint32_t k = (uint32_t)__ffs((int)m) - 1u;
          m &= m - 1u;
          // step 2: the matched arc and its destination's range: both lines are in the scalar cache now
          typedef uint32_t sv4 __attribute__((ext_vector_type(4)));
          typedef uint32_t sv2 __attribute__((ext_vector_type(2)));
          sv4 a;
          sv2 nx;
          const uint32_t o16 = k * 16u, o8 = k * 8u;
          asm volatile("s_load_dwordx4 %0, %2, %4\n\ts_load_dwordx2 %1, %3, %5\n\ts_waitcnt lgkmcnt(0)"
                       : "=&s"(a), "=&s"(nx)
                       : "s"(ap), "s"(np), "s"(o16), "s"(o8)
                       : "memory");
          if (!on_match(0u, fd, a[3], a[1], nx[0], nx[1], __uint_as_float(a[2]))) {
            ok = false;
            break;
          }
        }
      }
    }
    for (uint32_t f = 0; f < F && ok && !took_scalar; ++f) {
      const uint32_t fb = rl(nb, f), fc = rl(nc, f);
      const float fd = __uint_as_float(rl(__float_as_uint(d), f));
      for (uint32_t base = 0; base < fc && ok; base += 64) {
        const bool have = base + lane < fc;
        uint4 a = make_uint4(0, 0, 0, 0);
        uint2 nx = make_uint2(0, 0);
        if (have) {
          a = ld_global16(f2.arcs + fb + base + lane);
          nx = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(f2.anext) + 8ull * (fb + base + lane));
        }
        uint64_t m = __ballot(have && a.x == label);  // the label-equal run of fst2's (sorted) block, in arc order
        n_arcs += (uint32_t)__popcll(m);
        while (m) {
          const uint32_t l = (uint32_t)__ffsll((unsigned long long)m) - 1u;
          m &= m - 1;
          if (!on_match(f, fd, rl(a.w, l), rl(a.y, l), rl(nx.x, l), rl(nx.y, l), __uint_as_float(rl(a.z, l)))) {
            ok = false;
            break;
          }
        }
      }
    }
    lo = hi;
    hi += n_new;
    F = n_new;
    q = nq;
    nb = nnb;
    nc = nnc;
    d = nd;
    if (F) level = pos + 1;
  }
  if (!ok) {
    res.status = ST_NOT_A_STRING_CASE;
    STR_RESULT();
    return;
  }
  res.n_states = hi;
  res.n_arcs = n_arcs;
  res.n_levels = level + 1u;
  // final states: only level L can be final (fst1's single final state); rho = rho1 (x) rho2 (compose_fst_op.rs:420-449)
  lds_handoff();
  unsigned long long best = KEY_INF;
  float my_fin = INF;
  if (F && level == L) {  // the whole string was consumed
    const float fin1 = __uint_as_float(ld_global16(f1.srec + L).z);
    if (lane < F) {
      const float fin2 = __uint_as_float(ld_global16(f2.srec + q).z);
      if (fin1 != INF && fin2 != INF) my_fin = wtimes(fin1, fin2);
      if (d < INF && my_fin < INF) {
        const float tot = (d + my_fin) + 0.0f;
        if (tot < INF) best = ((unsigned long long)enc_f32(tot) << 32) | (lo + lane);
      }
    }
  }
  best = wave_min_u64(best);
  if (best != KEY_INF) {
    const uint32_t fp = (uint32_t)best;
    res.has_path = 1;
    res.hops = L;
    res.final_weight = __uint_as_float(rl(__float_as_uint(my_fin), fp - lo));
    res.total = dec_f32((uint32_t)(best >> 32));
    if (lane == 0) {  // walk the parents back (LDS); state k of the walk is in level L - k
      uint32_t cur = fp;
      for (uint32_t k = 0; k < L; ++k) {
        s_pth[k] = cur;
        cur = s_par[cur];
      }
    }
    lds_handoff();
    uint32_t poff = 0;
    if (lane == 0) poff = atomicAdd(path_cursor, L);
    poff = __shfl(poff, 0);
    if ((uint64_t)poff + L > path_cap) {
      res.status = ST_OVERFLOW_PATH;
    } else {
      res.path_off = poff;
      uint32_t facts = 0;
      for (uint32_t k = lane; k < L; k += 64) {  // arc k enters the k-th state created by the backtrace (shortest_path.rs:257-272)
        const uint32_t st = s_pth[k];
        const uint32_t il = ld_global16(f1.arcs + (L - 1u - k)).x;
        const uint32_t ol = s_ol[st];
        const float w = s_w[st];
        store_arc(path_buf + poff + k, il, ol, w, k);
        // fst_props.h path_arc_facts (is_zero / is_one: the reference's approximate ==, semiring.rs:68-73,159-168)
        const bool w_zero = w <= INF + props::KDELTA && INF <= w + props::KDELTA;
        const bool w_one = w <= props::KDELTA && 0.0f <= w + props::KDELTA;
        facts |= (il != ol ? 1u : 0u) | (il == WFST_EPS_LABEL ? 2u : 0u) | (il == WFST_EPS_LABEL && ol == WFST_EPS_LABEL ? 4u : 0u) |
                 (ol == WFST_EPS_LABEL ? 8u : 0u) | (!w_zero && !w_one ? 64u : 0u) | 128u;
      }
      for (int d = 32; d >= 1; d >>= 1) facts |= __shfl_xor(facts, d);
      res.facts = facts;
    }
  }
  STR_RESULT();
}

#undef STR_RESULT

// ---------------------------------------------------------------- host side
// ComposeFstOp::match_type (compose_fst_op.rs:169-197) with SortedMatcher::match_type
// (matchers/sorted_matcher.rs:56-85): decided from the property bits only.
uint32_t decide_match_mode(uint64_t p1, uint64_t p2) {
  using namespace props;
  const bool o_sorted = p1 & O_LABEL_SORTED, o_known = p1 & (O_LABEL_SORTED | NOT_O_LABEL_SORTED);
  const bool i_sorted = p2 & I_LABEL_SORTED, i_known = p2 & (I_LABEL_SORTED | NOT_I_LABEL_SORTED);
  if (o_sorted && i_sorted) return MODE_BOTH;
  if (o_sorted) return MODE_OUTPUT;
  if (i_sorted) return MODE_INPUT;
  // neither known-sorted: match_type(true) -> properties_check fails when the bit pair is unknown
  if (!o_known) throw Error("ComposeFst: 1st argument: label-sortedness properties are not known (sort?)");
  if (!i_known) throw Error("ComposeFst: 2nd argument: label-sortedness properties are not known (sort?)");
  throw Error(
      "ComposeFst: 1st argument cannot match on output labels and 2nd argument cannot match on input labels (sort?).");
}

FstView view_of(const wfst_fst* f) {
  FstView v;
  v.offsets = f->dev.offsets;
  v.arcs = f->dev.arcs;
  v.finals = f->dev.finals;
  v.noeps = f->dev.noeps;
  v.srec = f->dev.srec;
  v.anext = nullptr;
  v.n_states = f->n_states;
  v.start = (int32_t)f->start;
  v.n_arcs = f->n_arcs;
  return v;
}

uint32_t next_pow2(uint64_t x) {
  uint64_t p = 1;
  while (p < x) p <<= 1;
  if (p > 0x80000000ull) throw Error("composition too large for the wave-per-problem path");
  return (uint32_t)p;
}

Caps make_caps(uint64_t est_states, uint64_t est_arcs) {
  Caps c;
  uint64_t S = std::max<uint64_t>(est_states, 64);
  uint64_t A = std::max<uint64_t>(est_arcs, 128);
  if (S > 0x7FFFFFF0ull || A > 0x7FFFFFF0ull) throw Error("composition too large for the wave-per-problem path");
  c.S = (uint32_t)S;
  c.A = (uint32_t)A;
  c.H = next_pow2(2 * S + 128);
  c.W = 0;
  return c;
}

struct BatchRun {
  std::vector<Result> results;
  DBuf<char> arena;
  size_t stride = 0;
  Caps caps{};
  DBuf<wfst_tr> paths;
  std::vector<wfst_tr> h_paths;
  // in flight between launch_begin and launch_end
  size_t n = 0;
  bool want_paths = false;
  uint32_t path_cap = 1;
  DBuf<ProblemDesc> d_desc;
  DBuf<Result> d_res;
  DBuf<uint32_t> d_cursor;
  Result* h_res = nullptr;      // pinned (ctx->pinned_big): one batch in flight per context
  uint32_t* h_cursor = nullptr;
  // the first `eager` path records are copied to pinned memory right behind the kernel (no second round trip when
  // the paths fit, which they do unless T has input-epsilon cycles that stretch a path beyond its acceptor)
  wfst_tr* h_eager = nullptr;
  uint32_t eager = 0;
  const wfst_tr* host_paths = nullptr;
  bool string_kernel = false;  // this run went through string_compose_sp_kernel
  uint32_t done_ticket = 0;    // ... with results in pinned memory: what every Result::done holds when its problem is finished
  // descriptors read from, results and path arcs written to pinned host memory by the kernel itself: no copy commands
  // (each is a ~3 us API call on the host and a ~5 us command on the GPU) — when the whole path buffer fits there
  bool zero_copy = false;
};
constexpr uint32_t ZERO_COPY_ARCS = 1u << 18;  // 4 MB of pinned memory per run
// ... of the string o T kernel, whose path buffer is exact (a path through A_i o T has |A_i| - 1 arcs at most when T has no
// input epsilons): 64 MB of pinned memory — a batch of 4096 strings of 200 labels writes its 13 MB of path arcs to the host
// while it runs, instead of a copy command behind the kernel
constexpr uint32_t ZERO_COPY_ARCS_STRING = 1u << 22;

inline size_t run_pinned_bytes(size_t n, uint32_t eager) {
  return ((n * (sizeof(ProblemDesc) + sizeof(Result)) + 64 + (size_t)eager * sizeof(wfst_tr)) + 255) & ~(size_t)255;
}

// Enqueues descriptors, the kernel and the result copies on ctx's stream and returns without waiting.
// `pinned` (optional): where this run's staging lives inside ctx->pinned_big (two runs of one job share the buffer);
// `string_kernel`: launch string_compose_sp_kernel (no arena) instead of compose_wave_kernel.
template <uint32_t FLAGS>
void launch_begin(wfst_ctx* ctx, const std::vector<ProblemDesc>& descs, const FstView& f2, const Caps& caps, BatchRun& run,
                  bool want_paths, uint32_t eager_paths = 0, char* pinned = nullptr, bool string_kernel = false,
                  uint32_t max_f1_states = 0, uint32_t exact_path_cap = 0) {
  const size_t n = descs.size();
  DevicePool& pool = *ctx->pool;
  hipStream_t st = ctx->stream;
  run.n = n;
  run.want_paths = want_paths;
  run.caps = caps;
  run.string_kernel = string_kernel;
  run.stride = string_kernel ? 0 : arena_bytes(caps);
  run.arena = DBuf<char>(pool, string_kernel ? 256 : run.stride * n);
  run.d_desc = DBuf<ProblemDesc>(pool, n);
  run.d_res = DBuf<Result>(pool, n);
  run.d_cursor = DBuf<uint32_t>(pool, 1);
  uint32_t path_cap = want_paths ? (uint32_t)std::min<uint64_t>((uint64_t)caps.S * n, 0x7FFFFFFFull) : 1u;
  if (exact_path_cap && want_paths) path_cap = std::min(path_cap, exact_path_cap);  // (see ZERO_COPY_ARCS_STRING)
  run.path_cap = path_cap;
  run.eager = want_paths ? std::min(eager_paths, path_cap) : 0u;
  run.zero_copy = want_paths && pinned && run.eager == path_cap && !ctx->profiling;
  run.paths = DBuf<wfst_tr>(pool, run.zero_copy ? 1 : path_cap);
  ProblemDesc* h_desc = (ProblemDesc*)(pinned ? pinned : (char*)ctx->pinned_big.get(run_pinned_bytes(n, run.eager)));
  std::memcpy(h_desc, descs.data(), n * sizeof(ProblemDesc));
  run.h_res = (Result*)((char*)h_desc + n * sizeof(ProblemDesc));
  run.h_cursor = (uint32_t*)((char*)run.h_res + n * sizeof(Result));
  run.h_eager = (wfst_tr*)((char*)run.h_cursor + 64);
  const ProblemDesc* k_desc = run.zero_copy ? h_desc : run.d_desc.p;
  Result* k_res = run.zero_copy ? run.h_res : run.d_res.p;
  wfst_tr* k_paths = run.zero_copy ? run.h_eager : run.paths.p;
  if (!run.zero_copy) HIP_CHECK(hipMemcpyAsync(run.d_desc.p, h_desc, n * sizeof(ProblemDesc), hipMemcpyHostToDevice, st));
  HIP_CHECK(hipMemsetAsync(run.d_cursor.p, 0, sizeof(uint32_t), st));
  if (ctx->profiling) HIP_CHECK(hipEventRecord(ctx->ev0, st));
  if (string_kernel) {
    // waves per workgroup x states per problem: 64 KB of LDS per workgroup at most
    uint32_t wpb = 1, maxs = STR_MAXS;
    if (!std::getenv("WFST_STRING_UNPACKED") && n >= 16) {
      // a string of L arcs against an input-deterministic-ish T composes to a little more than L states; a problem that
      // outgrows its slice reports ST_NOT_A_STRING_CASE and is redone by the general kernel like any other misfit
      const uint64_t need = max_f1_states ? 2ull * max_f1_states + 64 : STR_MAXS;
      maxs = need <= 512 ? 512u : (need <= 1024 ? 1024u : STR_MAXS);
      wpb = (64u << 10) / (16u * maxs);
    }
    // tickets for serving-size batches only: the fence in front of a ticket writes the L2 back, and thousands of waves doing
    // that made a 4096-problem batch 0.7 ms slower; a kernel of milliseconds does not care about 10 us of wake-up latency
    static std::atomic<uint32_t> tickets{0};
    run.done_ticket = 0;
    if (run.zero_copy && n <= 1024 && !ctx->profiling && !std::getenv("WFST_BATCH_STREAM_WAIT")) {
      do run.done_ticket = tickets.fetch_add(1, std::memory_order_relaxed) + 1u; while (run.done_ticket == 0u);
      for (size_t i = 0; i < n; ++i) run.h_res[i].done = 0u;
    }
    string_compose_sp_kernel<<<(uint32_t)((n + wpb - 1) / wpb), 64 * wpb, (size_t)wpb * 16u * maxs, st>>>(
        k_desc, f2, k_res, k_paths, path_cap, run.d_cursor.p, (uint32_t)n, maxs, f2.n_arcs,
        // scalar arc-block loads where a wave is alone on its SIMD (a handful of strings): -7 % per level; with eight waves
        // per compute unit the other waves hide the vector latency anyway and the extra scalar instructions cost 3 %
        std::getenv("WFST_STRING_SCALAR") ? (uint32_t)std::atoi(std::getenv("WFST_STRING_SCALAR")) : (n <= 8 ? 1u : 0u), run.done_ticket);
  } else
    compose_wave_kernel<FLAGS><<<(uint32_t)n, 64, 0, st>>>(k_desc, f2, caps, run.arena.p, run.stride, k_res, k_paths, path_cap,
                                                            run.d_cursor.p);
  HIP_CHECK(hipGetLastError());
  if (ctx->profiling) HIP_CHECK(hipEventRecord(ctx->ev1, st));
  if (run.zero_copy) return;
  HIP_CHECK(hipMemcpyAsync(run.h_res, run.d_res.p, n * sizeof(Result), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipMemcpyAsync(run.h_cursor, run.d_cursor.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  if (run.eager)
    HIP_CHECK(hipMemcpyAsync(run.h_eager, run.paths.p, (size_t)run.eager * sizeof(wfst_tr), hipMemcpyDeviceToHost, st));
}

// Waits for the batch enqueued by launch_begin and brings results (and the path arcs) to the host.
void launch_end(wfst_ctx* ctx, BatchRun& run) {
  const size_t n = run.n;
  hipStream_t st = ctx->stream;
  Result* h_res = run.h_res;
  uint32_t* h_cursor = run.h_cursor;
  const bool want_paths = run.want_paths;
  const uint32_t path_cap = run.path_cap;
  // The string kernel's results in pinned memory carry a ticket each: the host reads them as they land — a stream wait first
  // retires whatever else has finished on the stream and wakes up ~10 us after the kernel — and falls back to the stream.
  bool seen = false;
  if (run.string_kernel && run.done_ticket != 0u) {
    const auto t0 = std::chrono::steady_clock::now();
    size_t next = 0;
    for (uint32_t spins = 0;; ++spins) {
      while (next < n && ((const volatile Result*)h_res)[next].done == run.done_ticket) ++next;
      if (next == n) {
        std::atomic_thread_fence(std::memory_order_acquire);
        seen = true;
        break;
      }
      cpu_relax();
      if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;  // (a long batch: sleep on the stream)
    }
  }
  if (!seen) HIP_CHECK(hipStreamSynchronize(st));
  else HIP_CHECK(hipPeekAtLastError());  // (no stream wait on this path.  Reports launch-configuration / sticky API errors the runtime has already
                                              // seen — NOT an asynchronous kernel fault: a chain that faults never writes its ticket, and the spin above then
                                              // ends in the stream wait, which reports it.  Peek, not Get: the error may belong to another thread's launch)
  if (ctx->profiling) {
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->stats.compose_ms = ms;
  }
  run.results.assign(h_res, h_res + n);
#ifdef WFST_PHASE_TIMING
  if (!run.string_kernel) {
    unsigned long long tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < n; ++i)
      for (int k = 0; k < 8; ++k) tot[k] += h_res[i].dbg[k];
    std::fprintf(stderr, "[phase cycles/problem] clear %llu | expand %llu | dedupe %llu | hash %llu | store+compact %llu | general %llu | epilogue %llu  (n=%zu, levels of problem 0: %u)\n",
                 tot[0] / n, tot[1] / n, tot[2] / n, tot[3] / n, tot[4] / n, tot[5] / n, tot[6] / n, n, h_res[0].n_levels);
  }
#endif
  if (want_paths && run.zero_copy) {
    run.host_paths = run.h_eager;  // written by the kernel
  } else if (want_paths) {
    const uint32_t used = std::min<uint32_t>(*h_cursor, path_cap);
    if (used <= run.eager) {
      run.host_paths = run.h_eager;
    } else {
      run.h_paths.resize(used);
      HIP_CHECK(hipMemcpyAsync(run.h_paths.data(), run.paths.p, (size_t)used * sizeof(wfst_tr), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      run.host_paths = run.h_paths.data();
    }
  }
}

template <uint32_t FLAGS>
void launch(wfst_ctx* ctx, const std::vector<ProblemDesc>& descs, const FstView& f2, const Caps& caps, BatchRun& run,
            bool want_paths) {
  launch_begin<FLAGS>(ctx, descs, f2, caps, run, want_paths);
  launch_end(ctx, run);
}

// builds the reference's linear shortest-path FST (see sssp.hip::build_path_fst)
wfst_fst* path_to_fst(wfst_ctx* ctx, const Result& r, const wfst_tr* path_arcs) {
  HostCsr h;
  uint32_t n_states = 0;
  int64_t start = -1;
  if (r.has_path) {
    n_states = r.hops + 1;
    h.finals.assign(n_states, INF);
    h.finals[0] = r.final_weight;
    if (r.hops) h.arcs.assign(path_arcs, path_arcs + r.hops);
    h.offsets.resize((size_t)n_states + 1);
    h.offsets[0] = 0;
    for (uint32_t k = 0; k <= r.hops; ++k) h.offsets[k + 1] = k;
    start = r.hops;
  } else {
    h.offsets.push_back(0);
  }
  const uint64_t p = r.has_path && r.facts != props::PATH_FACTS_NONE
                         ? props::linear_path_props_from_facts(true, r.hops, r.final_weight, r.facts)
                         : props::linear_path_props(r.has_path != 0, r.hops, r.final_weight, path_arcs);
  return make_host_fst(ctx, n_states, start, p, std::move(h));
}

// ... the same FST as a view: the arcs stay where the kernel wrote them (the job's pinned block, kept alive by the handle)
// until somebody reads the FST's arrays (ensure_host).  A batch of 64 costs ~8 us this way instead of ~47 (three vectors,
// 5 KB written and 3 KB copied per path), behind the kernel, on the critical path of a serving loop.
wfst_fst* path_view_fst(wfst_ctx* ctx, const Result& r, const wfst_tr* path_arcs, const std::shared_ptr<PinnedBlock>& block) {
  auto f = std::make_unique<wfst_fst>();
  f->ctx = ctx;
  f->owner_pool = ctx->pool;
  f->device = ctx->device;
  f->n_states = r.hops + 1;
  f->n_arcs = r.hops;
  f->start = r.hops;
  f->props = (r.facts != props::PATH_FACTS_NONE ? props::linear_path_props_from_facts(true, r.hops, r.final_weight, r.facts)
                                                : props::linear_path_props(true, r.hops, r.final_weight, path_arcs)) & props::ALL;
  f->path_block = block;
  f->path_arcs = path_arcs;
  f->path_final = r.final_weight;
  f->path_form = true;
  return f.release();
}

const char* status_name(uint32_t s) {
  switch (s) {
    case ST_OVERFLOW_STATES: return "states";
    case ST_OVERFLOW_ARCS: return "arcs";
    case ST_OVERFLOW_HASH: return "hash";
    case ST_OVERFLOW_PATH: return "path";
    default: return "ok";
  }
}

}  // namespace

wfst_fst* compose(wfst_ctx* ctx, const wfst_fst* f1, const wfst_fst* f2, bool connect, uint32_t filter) {
  if (f1->n_states > TUPLE_S1_MASK) throw Error("compose: the 1st FST has more than 2^30 states");
  const uint32_t mode = decide_match_mode(f1->props, f2->props);
  ensure_device(const_cast<wfst_fst*>(f1));
  ensure_device(const_cast<wfst_fst*>(f2));
  const bool has_start = f1->start >= 0 && f2->start >= 0;
  const uint64_t out_props = props::compose_result(f1->props, f2->props, connect, has_start);
  if (!has_start) {  // compute_start -> None: empty FST (lazy_fst.rs:229-232)
    HostCsr h;
    h.offsets.push_back(0);
    return make_host_fst(ctx, 0, -1, out_props, std::move(h));
  }
  std::vector<ProblemDesc> descs(1);
  descs[0].f1 = view_of(f1);
  descs[0].mode = mode;
  descs[0].filter = filter;
  const FstView v2 = view_of(f2);
  // first guess from the SMALLER operand (acceptor o transducer lattices are a few times the acceptor; the arena and its
  // hash table are cleared per attempt, so a guess from a 5M-state operand costs tens of ms before any work): larger
  // results retry with 4x
  uint64_t est_s = 4ull * std::max<uint64_t>(std::min(f1->n_states, f2->n_states), 64) + 1024;
  uint64_t est_a = 4ull * est_s;
  bool wide_ok = true;  // WFST_COMPOSE_PATH=wave pins the wave-per-problem kernel, =wide the wide driver (tests)
  if (const char* e = std::getenv("WFST_COMPOSE_PATH")) {
    if (std::strcmp(e, "wave") == 0) wide_ok = false;
    if (std::strcmp(e, "wide") == 0) return compose_wide(ctx, f1, f2, mode, filter, connect, out_props, est_s);
  }
  for (int attempt = 0;; ++attempt) {
    Caps caps = make_caps(est_s, est_a);
    if (wide_ok) caps.W = WIDE_COMPOSE_WIDTH;
    BatchRun run;
    if (connect)
      launch<FLAG_TRIM>(ctx, descs, v2, caps, run, false);
    else
      launch<0>(ctx, descs, v2, caps, run, false);
    const Result& r = run.results[0];
    ctx->stats.compose_states = r.n_states;
    ctx->stats.compose_arcs = r.n_arcs;
    if (r.status == ST_OK) {
      // carve the finished arrays out of the problem arena (same carve as the kernel)
      const Caps& c = run.caps;
      char* p = run.arena.p;
      p += al16((size_t)c.S * 8) + al16((size_t)c.H * 8) + al16((size_t)c.S * 8) + al16((size_t)c.S * 8);
      const wfst_tr* arcs = (const wfst_tr*)p; p += al16((size_t)c.A * 16);
      const wfst_tr* t_arcs = (const wfst_tr*)p; p += al16((size_t)c.A * 16);
      p += al16((size_t)c.H * 4);
      const uint32_t* off = (const uint32_t*)p; p += al16((size_t)(c.S + 1) * 4);
      const uint32_t* t_off = (const uint32_t*)p; p += al16((size_t)(c.S + 1) * 4);
      const float* fin = (const float*)p; p += al16((size_t)c.S * 4);
      const float* t_fin = (const float*)p;
      if (connect)
        return adopt_device(ctx, r.t_states, r.t_arcs, r.t_start, out_props, t_off, t_arcs, t_fin);
      return adopt_device(ctx, r.n_states, r.n_arcs, r.n_states ? 0 : -1, out_props, off, arcs, fin);
    }
    if (r.status == ST_SWITCH_WIDE) return compose_wide(ctx, f1, f2, mode, filter, connect, out_props, 4 * est_s);
    ctx->stats.compose_retries++;
    if (attempt > 24) throw Error(std::string("compose: arena overflow (") + status_name(r.status) + ") after retries");
    est_s *= 4;
    est_a *= 4;
    // a result that has outgrown WIDE_COMPOSE_STATES is not a job for ONE wave: the wide driver (compose_wide.hip: one
    // wave per composed state of a BFS level) does a million states in tens of ms where this kernel needs seconds
    if (wide_ok && est_s > WIDE_COMPOSE_STATES) return compose_wide(ctx, f1, f2, mode, filter, connect, out_props, est_s);
  }
}

}  // namespace wfst

// One fused batch in flight: begin enqueues everything on the context's stream and returns; end waits,
// re-runs the (rare) problems whose arena overflowed with 4x capacities, and assembles the path FSTs.
struct wfst_batch_job {
  wfst_ctx* ctx = nullptr;
  size_t n = 0;
  wfst::FstView v2{};
  std::vector<wfst::ProblemDesc> descs;
  std::vector<const wfst_fst*> accs;  // the operands (borrowed): problems the fused kernel hands back (ST_TIE_ORDER)
  const wfst_fst* t = nullptr;
  uint32_t filter = 0;
  std::vector<size_t> slow;           // ... are redone as compose + shortest_path at the end
  std::vector<size_t> todo;
  uint64_t est_s = 0, est_a = 0;
  wfst::BatchRun run;
  // problems that took the string o T kernel (fst1 a linear epsilon-free acceptor, fst2 without input epsilons)
  std::vector<size_t> todo_s;
  wfst::BatchRun run_s;
  wfst::FstView v2_s{};  // fst2 with its per-arc destination ranges
  std::shared_ptr<wfst::PinnedBlock> pin_block;  // descriptors, results and path arcs of the job's runs (pinned host memory)
};

namespace wfst {

wfst_batch_job* compose_shortest_path_batch_begin(wfst_ctx* ctx, const wfst_fst* const* accs, size_t n, const wfst_fst* t,
                                                  uint32_t filter) {
  auto job = std::make_unique<wfst_batch_job>();
  job->ctx = ctx;
  job->n = n;
  if (n == 0) return job.release();
  ensure_device(const_cast<wfst_fst*>(t));
  job->v2 = view_of(t);
  job->accs.assign(accs, accs + n);
  job->t = t;
  job->filter = filter;
  job->descs.resize(n);
  job->todo.resize(n);
  uint64_t max_states = 64;
  for (size_t i = 0; i < n; ++i) {
    if (!accs[i]) throw Error("null acceptor in batch");
    if (accs[i]->n_states > TUPLE_S1_MASK) throw Error("compose: a 1st FST has more than 2^30 states");
    job->descs[i].mode = decide_match_mode(accs[i]->props, t->props);
    ensure_device(const_cast<wfst_fst*>(accs[i]));
    job->descs[i].f1 = view_of(accs[i]);
    job->descs[i].filter = filter;
    job->todo[i] = i;
    max_states = std::max<uint64_t>(max_states, accs[i]->n_states);
  }
  job->est_s = 4ull * max_states + 256;
  job->est_a = 2ull * job->est_s;
  // which problems are "string o T" (the decoding case)?
  bool string_ok = (filter == 0 || filter == 3) && t->n_arcs > 0 && t->n_arcs < 0xFFFFFFFFull;
  if (const char* e = std::getenv("WFST_STRING_KERNEL")) string_ok = string_ok && std::atoi(e) != 0;
  if (string_ok) {
    bool any = false;
    for (size_t i = 0; i < n && !any; ++i) any = accs[i]->is_string && accs[i]->n_states <= STR_MAXS;
    string_ok = any && !has_input_epsilons(ctx, t);
  }
  std::vector<ProblemDesc> d_str, d_gen;
  std::vector<size_t> todo_gen;
  uint64_t eager_s = 0, eager_g = 0;  // a path through A_i o T has at most |A_i| - 1 arcs unless T loops on input epsilons
  uint32_t max_str_states = 0;
  if (string_ok) {
    job->v2_s = job->v2;
    job->v2_s.anext = ensure_anext(ctx, t);
    string_ok = job->v2_s.anext != nullptr;
  }
  for (size_t i = 0; i < n; ++i) {
    if (string_ok && accs[i]->is_string && accs[i]->n_states <= STR_MAXS) {
      job->todo_s.push_back(i);
      d_str.push_back(job->descs[i]);
      eager_s += accs[i]->n_states;
      max_str_states = std::max(max_str_states, accs[i]->n_states);
    } else {
      todo_gen.push_back(i);
      d_gen.push_back(job->descs[i]);
      eager_g += accs[i]->n_states;
    }
  }
  job->todo.swap(todo_gen);
  uint32_t e_s = (uint32_t)std::min<uint64_t>(eager_s, 1u << 22), e_g = (uint32_t)std::min<uint64_t>(eager_g, 1u << 22);
  {  // a run whose whole path buffer fits in pinned memory writes there itself (launch_begin: zero_copy)
    const Caps c0 = make_caps(job->est_s, job->est_a);
    const uint64_t cap_s = (uint64_t)c0.S * d_str.size(), cap_g = (uint64_t)c0.S * d_gen.size();
    const bool allow = !std::getenv("WFST_BATCH_COPY");  // tests: the copy-command path
    if (allow && cap_s && cap_s <= ZERO_COPY_ARCS) e_s = (uint32_t)cap_s;
    else if (allow && eager_s && eager_s <= ZERO_COPY_ARCS_STRING) e_s = (uint32_t)eager_s;  // = the run's whole path buffer
    if (allow && cap_g && cap_g <= ZERO_COPY_ARCS) e_g = (uint32_t)cap_g;
  }
  const size_t pin_s = d_str.empty() ? 0 : run_pinned_bytes(d_str.size(), e_s);
  const size_t pin_g = d_gen.empty() ? 0 : run_pinned_bytes(d_gen.size(), e_g);
  // (a block of the context's ring, not its one staging buffer: the path FSTs this job returns keep pointing into it)
  job->pin_block = ctx->pinned_ring->take(pin_s + pin_g + 256);
  char* pin = (char*)job->pin_block->p;
  if (!d_str.empty())
    launch_begin<FLAG_SP>(ctx, d_str, job->v2_s, make_caps(job->est_s, job->est_a), job->run_s, true, e_s, pin, true, max_str_states,
                          (uint32_t)std::min<uint64_t>(eager_s, 0x7FFFFFFFull));
  if (!d_gen.empty())
    launch_begin<FLAG_SP>(ctx, d_gen, job->v2, make_caps(job->est_s, job->est_a), job->run, true, e_g, pin + pin_s, false);
  return job.release();
}

void compose_shortest_path_batch_end(wfst_batch_job* job_raw, wfst_fst** outs, uint64_t* composed_arcs, const PackedSink* sink) {
  std::unique_ptr<wfst_batch_job> job(job_raw);  // consumed whatever happens
  wfst_ctx* ctx = job->ctx;
  const size_t n = job->n;
  if (composed_arcs) *composed_arcs = 0;
  if (n == 0) return;
  if (!sink)
    for (size_t i = 0; i < n; ++i) outs[i] = nullptr;
  // result i: a path FST handle, or (sink) its record — straight from the kernel's result and path buffers
  const size_t rec_words = sink ? 4 + 4 * (size_t)sink->max_arcs : 0;
  // (views only into blocks of serving size: one surviving result would otherwise keep tens of MB pinned)
  const bool views = job->pin_block && job->pin_block->cap <= (4u << 20) && !std::getenv("WFST_BATCH_EAGER_PATHS");
  auto emit = [&](size_t i, const Result& r, const wfst_tr* arcs, bool in_block) {
    if (sink) pack_path_record(sink->out + i * rec_words, sink->max_arcs, r.has_path != 0, r.hops, r.final_weight, arcs);
    else if (views && in_block && r.has_path && r.hops) outs[i] = path_view_fst(ctx, r, arcs, job->pin_block);
    else outs[i] = path_to_fst(ctx, r, arcs);
  };
  uint64_t tot_arcs = 0, tot_states = 0, n_string_ok = 0;
  double ms = 0;
  const bool timing = std::getenv("WFST_HOST_TIMING") != nullptr;
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto t_a = tnow();
  try {
    if (!job->todo_s.empty()) {  // results of the string o T kernel; what it did not cover joins the general list
      launch_end(ctx, job->run_s);
      const auto t_w = tnow();
      ms += ctx->stats.compose_ms;
      std::vector<size_t> more, ok;
      for (size_t k = 0; k < job->todo_s.size(); ++k) {
        const Result& r = job->run_s.results[k];
        if (r.status != ST_OK) {
          (r.status == ST_TIE_ORDER ? job->slow : more).push_back(job->todo_s[k]);
          continue;
        }
        tot_arcs += r.n_arcs;
        tot_states += r.n_states;
        n_string_ok += 1;
        ok.push_back(k);
      }
      // the path FSTs (three small vectors and a handle each: ~0.4 us): a large batch shares them out among host threads
      // (records are a copy of ~3 KB each: fewer threads than for handles — creating a thread costs as much as ~50 records)
      const unsigned n_thr = std::getenv("WFST_HOST_THREADS")
                                 ? host_threads(ok.size())
                                 : (unsigned)std::min<size_t>(std::min(sink ? 8u : 16u, host_threads(1u << 16)), ok.size() / (sink ? 512 : 256));
      parallel_chunks(n_thr, ok.size(), 64, [&](unsigned, uint64_t b, uint64_t e) {
        for (uint64_t q = b; q < e; ++q) {
          const Result& r = job->run_s.results[ok[q]];
          emit(job->todo_s[ok[q]], r, r.has_path && r.hops ? job->run_s.host_paths + r.path_off : nullptr, job->run_s.zero_copy);
        }
      });
      if (timing)
        std::fprintf(stderr, "[batch_end] string run: wait %.1f us, %zu results on %u thread(s) %.1f us\n",
                     std::chrono::duration<double, std::micro>(t_w - t_a).count(), ok.size(), std::max(1u, n_thr),
                     std::chrono::duration<double, std::micro>(tnow() - t_w).count());
      if (!more.empty()) {
        if (!job->todo.empty()) {  // the general run of this job is still in flight: collect it first
          launch_end(ctx, job->run);
          std::vector<size_t> again;
          for (size_t k = 0; k < job->todo.size(); ++k) {
            const Result& r = job->run.results[k];
            if (r.status != ST_OK) {
              (r.status == ST_TIE_ORDER ? job->slow : again).push_back(job->todo[k]);
              continue;
            }
            tot_arcs += r.n_arcs;
            tot_states += r.n_states;
            emit(job->todo[k], r, r.has_path && r.hops ? job->run.host_paths + r.path_off : nullptr, job->run.zero_copy && job->run.host_paths == job->run.h_eager);
          }
          more.insert(more.end(), again.begin(), again.end());
        }
        job->todo.swap(more);
        std::vector<ProblemDesc> cur(job->todo.size());
        for (size_t k = 0; k < job->todo.size(); ++k) cur[k] = job->descs[job->todo[k]];
        job->run = BatchRun{};
        launch_begin<FLAG_SP>(ctx, cur, job->v2, make_caps(job->est_s, job->est_a), job->run, true);
      }
    }
    for (int attempt = 0; !job->todo.empty(); ++attempt) {
      launch_end(ctx, job->run);
      auto t_b = tnow();
      if (timing) std::fprintf(stderr, "[batch_end] wait+copy %.1f us\n", std::chrono::duration<double, std::micro>(t_b - t_a).count());
      ms += ctx->stats.compose_ms;
      std::vector<size_t> again;
      for (size_t k = 0; k < job->todo.size(); ++k) {
        const Result& r = job->run.results[k];
        if (r.status != ST_OK) {
          (r.status == ST_TIE_ORDER ? job->slow : again).push_back(job->todo[k]);
          continue;
        }
        tot_arcs += r.n_arcs;
        tot_states += r.n_states;
        emit(job->todo[k], r, r.has_path && r.hops ? job->run.host_paths + r.path_off : nullptr, job->run.zero_copy && job->run.host_paths == job->run.h_eager);
      }
      if (timing) std::fprintf(stderr, "[batch_end] assemble %.1f us\n", std::chrono::duration<double, std::micro>(tnow() - t_b).count());
      job->todo.swap(again);
      if (job->todo.empty()) break;
      ctx->stats.compose_retries++;
      if (attempt > 24) throw Error("compose_shortest_path_batch: arena overflow after retries");
      job->est_s *= 4;
      job->est_a *= 4;
      std::vector<ProblemDesc> cur(job->todo.size());
      for (size_t k = 0; k < job->todo.size(); ++k) cur[k] = job->descs[job->todo[k]];
      job->run = BatchRun{};
      launch_begin<FLAG_SP>(ctx, cur, job->v2, make_caps(job->est_s, job->est_a), job->run, true);
    }
    for (size_t idx : job->slow) {  // handed back by the fused kernel: the two-step route (always applicable)
      std::unique_ptr<wfst_fst> c(compose(ctx, job->accs[idx], job->t, true, job->filter));
      tot_arcs += c->n_arcs;
      tot_states += c->n_states;
      if (sink) {
        std::unique_ptr<wfst_fst> p(shortest_path_n1(ctx, c.get()));
        pack_path_record(sink->out + idx * rec_words, sink->max_arcs, p.get());
      } else {
        outs[idx] = shortest_path_n1(ctx, c.get());
      }
    }
  } catch (...) {
    (void)hipStreamSynchronize(ctx->stream);  // (a run of this job may still be in flight: its buffers go with the job)
    if (!sink)
      for (size_t i = 0; i < n; ++i) {
        delete outs[i];
        outs[i] = nullptr;
      }
    throw;
  }
  ctx->stats.compose_states = tot_states;
  ctx->stats.compose_arcs = tot_arcs;
  ctx->stats.compose_ms = ms;
  ctx->stats.string_problems = n_string_ok;
  if (composed_arcs) *composed_arcs = tot_arcs;
}

wfst_ctx* batch_job_ctx(const wfst_batch_job* job) { return job->ctx; }

void compose_shortest_path_batch_abandon(wfst_batch_job* job) {
  if (!job) return;
  if (job->n) (void)hipStreamSynchronize(job->ctx->stream);  // the kernel may still be writing into the job's buffers
  delete job;
}

void compose_shortest_path_batch(wfst_ctx* ctx, const wfst_fst* const* accs, size_t n, const wfst_fst* t, bool /*connect*/,
                                 wfst_fst** outs, uint64_t* composed_arcs, uint32_t filter) {
  compose_shortest_path_batch_end(compose_shortest_path_batch_begin(ctx, accs, n, t, filter), outs, composed_arcs);
}

}  // namespace wfst
