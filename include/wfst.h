/*
 * wfst.h — C-ABI of the MI355X-native WFST compose + shortest-path engine (libwfst_amd.so).
 *
 * Drop-in boundary for rustfst's `algorithms::compose` and `algorithms::shortest_path` on
 * VectorFst<TropicalWeight>.  Conventions are the ones rustfst-ffi already uses, so a Rust shim
 *  can bind these with `extern "C"` exactly as it binds its own cdylib:
 *   - every call returns a status (0 = OK, 1 = KO)            rustfst-ffi/src/lib.rs:29-37
 *   - the error text is thread-local, fetched + freed by the caller
 *                                                               rustfst-ffi/src/lib.rs:39-85
 *   - objects are opaque heap handles written into an out-param and released by an explicit
 *     destroy; inputs are borrowed, outputs are owned by the caller
 *                                                               rustfst-ffi/src/algorithms/compose.rs:274-334
 *   - labels / state ids are `unsigned int`, an arc is CTr      rustfst-ffi/src/lib.rs:19-27, src/tr.rs:8-21
 * Plain pointers and sizes only; no torch / HIP types in any signature (a HIP stream crosses
 * as `void*`).
 *
 * Semantics are the reference's (file:line cited per entry point).  Two documented deviations,
 * both inside behaviour the reference itself leaves undefined or approximate:
 *   1. TropicalWeight `==` is exact here; the reference's is |a-b| <= 1/1024
 *      (rustfst/src/semirings/semiring.rs:159-168).  Identical results whenever weights lie on a
 *      grid coarser than 1/1024 (all fixtures / benchmarks); otherwise this engine returns the true
 *      (min,+) optimum.
 *   2. Among equal-weight shortest paths the winner is the canonical one (fewest arcs, then smallest
 *      (state,arc position) predecessor); the reference picks "first relaxer in queue order" and its
 *      own tests refuse to compare such outputs structurally
 *      (rustfst/src/tests_openfst/algorithms/shortest_path.rs:69-92).
 */
#ifndef WFST_AMD_H
#define WFST_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WFST_ABI_VERSION 7 /* 7: wfst_fst_set_start; 6: wfst_ctx_set_resident_share; 5: wfst_ctx_get_sweep_modes, relax_kernel may be 3, wfst_stats gained tied_choices;
                             * 2: wfst_stats gained relax_kernel; 3: wfst_comm_* / wfst_gather_paths_*, ..._batch_packed;
                             * 4: wfst_stats gained resident_aborts, relax_kernel may be 2; wfst_comm_create_host, wfst_gather_records_begin */

typedef enum { WFST_OK = 0, WFST_KO = 1 } wfst_status; /* RUSTFST_FFI_RESULT, rustfst-ffi/src/lib.rs:29-37 */

/* == rustfst-ffi CTr (src/tr.rs:8-21) == the 16-byte on-disk arc (parsers/bin_fst/utils_parsing.rs:28-44) */
typedef struct {
  uint32_t ilabel;
  uint32_t olabel;
  float weight; /* TropicalWeight: +inf = zero, 0.0 = one */
  uint32_t nextstate;
} wfst_tr;

#define WFST_EPS_LABEL 0u             /* rustfst/src/lib.rs:236 */
#define WFST_NO_LABEL 0xFFFFFFFFu     /* rustfst/src/lib.rs:292 */
#define WFST_NO_STATE_ID 0xFFFFFFFFu  /* rustfst/src/lib.rs:298 */

typedef struct wfst_ctx wfst_ctx; /* one per (host thread, GPU); owns a HIP stream + device pools */
typedef struct wfst_fst wfst_fst; /* an FST resident in HBM as CSR (and/or on the host for small results) */

/* ---- errors: rustfst_ffi_get_last_error / rustfst_destroy_string (rustfst-ffi/src/lib.rs:58-85) ---- */
wfst_status wfst_last_error(char** msg); /* takes the message (thread-local); caller frees */
wfst_status wfst_string_destroy(char* msg);
uint32_t wfst_abi_version(void);

/* ---- context ---- */
wfst_status wfst_ctx_create(int device, wfst_ctx** out);
/* use an existing HIP stream (e.g. torch.cuda.current_stream().cuda_stream); not owned */
wfst_status wfst_ctx_create_on_stream(int device, void* hip_stream, wfst_ctx** out);
/* context whose (own) stream may only use the compute units set in cu_mask (bit i of word i/32 = CU i;
 * hipExtStreamCreateWithCUMask).  For serving several request classes on one GPU: the fused batch is a handful of
 * long single-wave workgroups whose dependent loads slow down ~40 % when bandwidth-hungry kernels of another context
 * share their CUs; giving each context disjoint CUs removes that interference. */
wfst_status wfst_ctx_create_with_cu_mask(int device, const uint32_t* cu_mask, uint32_t mask_words, wfst_ctx** out);
wfst_status wfst_ctx_destroy(wfst_ctx* ctx);
wfst_status wfst_ctx_synchronize(wfst_ctx* ctx);
wfst_status wfst_ctx_stream(wfst_ctx* ctx, void** hip_stream);

/* ---- FST handles.  What the Rust shim produces by walking the trait surface
 * (start / num_states / get_trs / final_weight / properties: rustfst/src/fst_traits/fst.rs:18-226):
 *   offsets[n_states+1], arcs[offsets[n]] in per-state stored order, finals[n] with +inf = None
 *   (same sentinel as on disk, vector_fst/serializable_fst.rs:78-80), props = FstProperties bits
 *   (rustfst/src/fst_properties/properties.rs:22-103), start = -1 for None. ---- */
wfst_status wfst_fst_upload(wfst_ctx* ctx, uint32_t n_states, int64_t start, const uint32_t* offsets,
                            const wfst_tr* arcs, const float* finals, uint64_t props, wfst_fst** out);
/* same, but the three arrays already live in HBM on ctx's device (e.g. torch tensors); copied */
wfst_status wfst_fst_upload_device(wfst_ctx* ctx, uint32_t n_states, int64_t start, const uint32_t* d_offsets,
                                   const wfst_tr* d_arcs, const float* d_finals, uint64_t props, wfst_fst** out);
/* n FSTs in one shot (one device arena, one copy): arrays are concatenated; state_base[i] /
 * arc_base[i] give FST i's first state / arc; offsets are per-FST relative (each has n_i+1 entries,
 * concatenated: FST i's offsets start at state_base[i] + i). */
wfst_status wfst_fst_upload_many(wfst_ctx* ctx, size_t n, const uint32_t* n_states, const int64_t* starts,
                                 const uint32_t* offsets_cat, const wfst_tr* arcs_cat, const float* finals_cat,
                                 const uint64_t* props, wfst_fst** outs);
/* OpenFST binary, arc type "standard": vec_fst_from_bytes / vec_fst_to_bytes
 * (rustfst-ffi/src/fst/vector_fst.rs:319-354; format rustfst/src/parsers/bin_fst/fst_header.rs:71-112,
 * rustfst/src/fst_impls/vector_fst/serializable_fst.rs:45-168). Symbol tables are skipped.
 * The reader accepts fst_type "vector" (version >= 2) and "const" (version 1 = 16-byte aligned blocks, version 2;
 * rustfst/src/fst_impls/const_fst/serializable_fst.rs:176-237: const_fst_from_bytes, rustfst-ffi/src/fst/const_fst.rs).
 * _to_openfst_bytes writes "vector" v2, _to_openfst_const_bytes writes "const" v2 (ConstFst::store, :41-89). */
wfst_status wfst_fst_from_openfst_bytes(wfst_ctx* ctx, const uint8_t* data, size_t len, wfst_fst** out);
wfst_status wfst_fst_to_openfst_bytes(const wfst_fst* fst, uint8_t** data, size_t* len);
wfst_status wfst_fst_to_openfst_const_bytes(const wfst_fst* fst, uint8_t** data, size_t* len);
wfst_status wfst_bytes_destroy(uint8_t* data);

wfst_status wfst_fst_info(const wfst_fst* fst, uint32_t* n_states, uint64_t* n_arcs, int64_t* start,
                          uint64_t* props);
/* copy out: offsets[n_states+1], arcs[n_arcs], finals[n_states] (any pointer may be NULL) */
wfst_status wfst_fst_download(const wfst_fst* fst, uint32_t* offsets, wfst_tr* arcs, float* finals);
wfst_status wfst_fst_destroy(wfst_fst* fst);
/* destroys fsts[0..n) (NULL entries are skipped): the outs[] of a fused batch in one call */
wfst_status wfst_fst_destroy_many(wfst_fst* const* fsts, size_t n);

/* ---- compose: fst_compose / fst_compose_with_config (rustfst-ffi/src/algorithms/compose.rs:308-372)
 *      = rustfst::algorithms::compose::{compose, compose_with_config}
 *        (rustfst/src/algorithms/compose/compose_static.rs:166-306).
 * compose_filter uses the ffi numbering (compose.rs:20-33): 0 Auto (= Sequence, compose_fst.rs:58-92), 1 Null,
 * 2 Trivial, 3 Sequence, 4 AltSequence, 5 Match, 6 NoMatch — all with the default SortedMatcher pair
 * (rustfst/src/algorithms/compose/compose_filters/{null,trivial,sequence,alt_sequence,match,no_match}_compose_filter.rs);
 * custom matcher configs (sigma matcher) are not part of this ABI.  cfg == NULL means
 * ComposeConfig::default() = {Auto, connect = true} (compose_static.rs:56-65).
 * KO with the reference's message when neither side is known label-sorted
 * (compose/compose_fst_op.rs:169-197).  Output state ids / arc order are the reference's
 * (FIFO BFS discovery order, then stable trim: lazy/lazy_fst.rs:226-269, connect.rs:51-66). ---- */
typedef struct {
  uint32_t compose_filter;
  uint32_t connect; /* bool */
} wfst_compose_config;
wfst_status wfst_compose(wfst_ctx* ctx, const wfst_fst* fst1, const wfst_fst* fst2,
                         const wfst_compose_config* cfg, wfst_fst** out);

/* ---- shortest path: fst_shortest_path / fst_shortest_path_with_config
 *      (rustfst-ffi/src/algorithms/shortest_path.rs:44-83) = rustfst::algorithms::shortest_path
 *      (rustfst/src/algorithms/shortest_path.rs:76-133).  cfg == NULL means
 *      ShortestPathConfig::default() = {delta 1e-6, nshortest 1, unique false} (:31-39).
 *      nshortest == 0 -> empty FST (:118-120).  nshortest == 1 (unique ignored, as in the reference):
 *      relaxation + backtrace on the GPU; output = linear FST numbered backwards, state 0 final (:241-282).
 *      nshortest > 1, unique = false (:135-170): shortest_distance and reverse() on the GPU, the sequential
 *      n_shortest_path heap search (:409-518) + connect on the host; output = the reference's path tree.
 *      nshortest > 1 with unique = true (:157-165): the reversed FST is determinized first (determinize_with_distance,
 *      determinize/determinize_static.rs:24-39; host code, input must be an ACCEPTOR by its property word or the call is
 *      KO "DeterminizeFsaImpl : expected acceptor as argument", as in the reference), then the same search: the n best
 *      DISTINCT strings.  Where the reference keeps a weighted subset in HashMap order (unspecified, and part of a
 *      state's identity there) this library keeps it in ascending state order. ---- */
typedef struct {
  float delta;
  uint64_t nshortest;
  uint32_t unique; /* bool */
} wfst_shortest_path_config;
wfst_status wfst_shortest_path(wfst_ctx* ctx, const wfst_fst* fst, const wfst_shortest_path_config* cfg,
                               wfst_fst** out);
/* single-source (min,+) distances from the start state (what single_shortest_path computes into
 * `distance`, shortest_path.rs:173-239) copied to host arrays of n_states entries; hops may be NULL. */
wfst_status wfst_shortest_distance(wfst_ctx* ctx, const wfst_fst* fst, float* distance, uint32_t* hops);

/* ---- connect: fst_connect (rustfst-ffi/src/algorithms/connect.rs:14-23) = rustfst::algorithms::connect
 *      (rustfst/src/algorithms/connect.rs:51-66): the states that are accessible from the start state and can reach a
 *      final state, renumbered stably (del_states, vector_fst/mutable_fst.rs:132-189), arcs into deleted states dropped.
 *      The reference trims in place; here a NEW handle is returned (the caller destroys the old one). ---- */
wfst_status wfst_connect(wfst_ctx* ctx, const wfst_fst* fst, wfst_fst** out);

/* ---- rm_epsilon: fst_rm_epsilon (rustfst-ffi/src/algorithms/rm_epsilon.rs) = rustfst::algorithms::rm_epsilon
 *      (rustfst/src/algorithms/rm_epsilon/rm_epsilon_static.rs:50-163) with its default configuration (connect, no
 *      thresholds): every epsilon:epsilon arc removed, the weighted relation kept, the result connected.  The reference
 *      works in place; here a NEW handle is returned.  An FST without a start state is returned unchanged. ---- */
wfst_status wfst_rm_epsilon(wfst_ctx* ctx, const wfst_fst* fst, wfst_fst** out);

/* ---- project: fst_project (rustfst-ffi/src/algorithms/project.rs:45-70) = rustfst::algorithms::project
 *      (rustfst/src/algorithms/projection.rs:65-95), in place on the device-resident arcs.  project_output == 0:
 *      ProjectType::ProjectInput (olabel := ilabel), != 0: ProjectOutput (ilabel := olabel); the property word follows
 *      project_properties (fst_properties/mutate_properties.rs:365-445).  The usual recipe around the hot path is
 *      compose -> project -> shortest_path (rustfst/src/lib.rs:70-82). ---- */
wfst_status wfst_fst_project(wfst_ctx* ctx, wfst_fst* fst, int project_output);

/* ---- look-ahead composition: the configuration rustfst-cli/src/cmds/compose.rs:77-181 (ComposeType::LookAhead) and
 *      rustfst/src/tests_openfst/algorithms/compose.rs:118-254 build by hand — there is no single reference entry point:
 *        graph1look = MatcherFst::new_with_relabeling(fst1, &mut fst2, true)        (compose/matcher_fst.rs:73-94)
 *        M1 = LabelLookAheadMatcher<SortedMatcher> with OUTPUT_LOOKAHEAD_MATCHER | LOOKAHEAD_WEIGHT | LOOKAHEAD_PREFIX |
 *             LOOKAHEAD_EPSILONS | LOOKAHEAD_NON_EPSILON_PREFIX, M2 = SortedMatcher
 *        filter = PushLabels(PushWeights(LookAhead(AltSequence))) with SMatchOutput  (compose/lookahead_filters/...)
 *        ComposeFst::new_with_options(..).compute()                                   (no connect)
 *      wfst_lookahead_create  = MatcherFst::new on fst1: LabelReachable::compute_data(fst1, reach_input = false)
 *        (compose/label_reachable.rs:135-273, host) + fst1's olabels relabelled and re-sorted (:63-93); the relabelled FST
 *        and the per-state reachable-label intervals live in HBM.  KO "StateReachable: Final state contained in a cycle"
 *        like the reference (state_reachable.rs:62-64).
 *      wfst_lookahead_relabel = LabelLookAheadRelabeler::relabel(fst2, .., relabel_input = true) + tr_sort(ILabelCompare)
 *        (lookahead_matchers/label_lookahead_relabeler.rs:27-41, cmds/compose.rs:151): a NEW handle; labels fst1 never
 *        emits get fresh indices, the map inside `la` grows as the reference's does.
 *      wfst_compose_lookahead = the composition itself on the GPU; output numbered like LazyFst::compute, not connected.
 *      The handle owns the relabelled fst1 (wfst_lookahead_fst1 lends it). ---- */
typedef struct wfst_lookahead wfst_lookahead;
wfst_status wfst_lookahead_create(wfst_ctx* ctx, const wfst_fst* fst1, wfst_lookahead** out);
wfst_status wfst_lookahead_relabel(wfst_lookahead* la, const wfst_fst* fst2, wfst_fst** out);
wfst_status wfst_lookahead_fst1(const wfst_lookahead* la, const wfst_fst** out);
wfst_status wfst_compose_lookahead(wfst_ctx* ctx, const wfst_lookahead* la, const wfst_fst* relabeled_fst2, wfst_fst** out);
/* n independent look-ahead compositions against the same first operand in ONE launch (one wavefront per problem, like
 * wfst_compose_shortest_path_batch); outs[i] == wfst_compose_lookahead(ctx, la, relabeled_fst2s[i]).  On KO no output is
 * left allocated. */
wfst_status wfst_compose_lookahead_batch(wfst_ctx* ctx, const wfst_lookahead* la, const wfst_fst* const* relabeled_fst2s,
                                         size_t n, wfst_fst** outs);
wfst_status wfst_lookahead_destroy(wfst_lookahead* la);
/* LabelReachableData of a look-ahead handle: sizes, then the arrays (interval_offsets[n_states + 1], intervals[2 *
 * n_intervals] = half-open [begin, end) pairs over relabelled labels, labels[n_labels] ascending with their indices;
 * the NO_LABEL entry is the final label).  Any output may be NULL. */
wfst_status wfst_lookahead_info(const wfst_lookahead* la, uint32_t* n_states, uint64_t* n_intervals, uint32_t* n_labels,
                                uint32_t* final_label);
wfst_status wfst_lookahead_download(const wfst_lookahead* la, uint32_t* interval_offsets, uint32_t* intervals,
                                    uint32_t* labels, uint32_t* indices);
/* host-only handle (no GPU, no relabelled FST): LabelReachable::compute_data on flat CSR arrays (offsets[n_states + 1],
 * arcs, finals with +inf = not final).  Serves wfst_lookahead_info / _download / _destroy only. */
wfst_status wfst_label_reachable_compute(uint32_t n_states, const uint32_t* offsets, const wfst_tr* arcs, const float* finals,
                                         int reach_input, wfst_lookahead** out);

/* asynchronous form of wfst_shortest_path for nshortest == 1: _begin queues the relaxation (and, from the second
 * query of an FST on, the final-state search, backtrace and read-back behind it) on ctx's stream and returns; _end
 * waits, continues the relaxation if it needed more sweeps than the previous query, and returns the same FST the
 * synchronous call returns, freeing the job (also on error; out == NULL abandons it).  One job in flight per context
 * and no other call on that context between _begin and _end; fst must stay alive until _end.  cfg as in
 * wfst_shortest_path; nshortest != 1 -> KO "unsupported". */
typedef struct wfst_sp_job wfst_sp_job;
wfst_status wfst_shortest_path_begin(wfst_ctx* ctx, const wfst_fst* fst, const wfst_shortest_path_config* cfg,
                                     wfst_sp_job** job);
wfst_status wfst_shortest_path_end(wfst_sp_job* job, wfst_fst** out);

/* asynchronous form of the fused batch: _begin enqueues the whole pipeline on ctx's stream and returns at once
 * (so that the caller can issue other work, e.g. wfst_shortest_path on ANOTHER context, which then overlaps on
 * the GPU); _end waits, fills outs[0..n) / composed_arcs exactly like the synchronous call and frees the job
 * (also on error).  One job in flight per context; acceptors and t must stay alive until _end.
 * The outs[] of a batch of up to a few thousand strings are views into the batch's pinned result block: complete FST
 * handles whose arrays are built on first access (download, an algorithm, a writer); see INTEGRATION.md. */
typedef struct wfst_batch_job wfst_batch_job;
wfst_status wfst_compose_shortest_path_batch_begin(wfst_ctx* ctx, const wfst_fst* const* acceptors, size_t n,
                                                   const wfst_fst* t, const wfst_compose_config* ccfg,
                                                   const wfst_shortest_path_config* scfg, wfst_batch_job** job);
wfst_status wfst_compose_shortest_path_batch_end(wfst_batch_job* job, wfst_fst** outs, uint64_t* composed_arcs);

/* ---- reverse (rustfst/src/algorithms/reverse.rs:33-87; FFI fst_reverse): state 0 of the result is a new super-initial
 *      state with one eps:eps arc per final state of fst (weight = its final weight), state s + 1 holds the arcs INTO s
 *      turned around, in (source state, arc position) order; start + 1 is final with weight one.  The transpose is built
 *      on the GPU and cached on the handle (the n > 1 shortest-path search uses the same one). ---- */
wfst_status wfst_reverse(wfst_ctx* ctx, const wfst_fst* fst, wfst_fst** out);

/* ---- tr_sort (rustfst/src/algorithms/tr_sort.rs:13-62; FFI fst_tr_sort, rustfst-ffi/src/algorithms/tr_sort.rs:15):
 *      in-place, stable, per-state sort of the device-resident arcs by ilabel (ilabel_cmp != 0, ILabelCompare)
 *      or olabel (OLabelCompare), followed by the reference's property update.  This is what makes an FST
 *      acceptable to wfst_compose (SortedMatcher needs the sorted bit). ---- */
wfst_status wfst_fst_tr_sort(wfst_ctx* ctx, wfst_fst* fst, int ilabel_cmp);

/* ---- set_start on a device-resident handle (MutableFst::set_start, rustfst/src/fst_impls/vector_fst/mutable_fst.rs:35-44;
 *      FFI vec_fst_set_start, rustfst-ffi/src/fst/vector_fst.rs:26-33): KO "The state {state} doesn't exist" for a state beyond
 *      the FST, otherwise the start state and the property word change as the reference's do (set_start_properties,
 *      fst_properties/mutate_properties.rs:7-13); the arcs stay where they are in HBM and everything cached on the handle that
 *      does not depend on the start state (region plan, transpose, packed arcs) is kept — a shortest_path query per source on
 *      one resident FST is this call + wfst_shortest_path (ABI 7).  Like every mutation of a handle: not while a query on it
 *      is in flight on any context. ---- */
wfst_status wfst_fst_set_start(wfst_ctx* ctx, wfst_fst* fst, uint32_t state);

/* ---- fused batch: for each acceptor i: shortest_path(compose(acceptors[i], t)) — the loop a
 * caller writes around the two reference entry points; here one device-resident pipeline.
 * outs[i] are small host-resident FSTs. composed_arcs (may be NULL) receives the total number of
 * arcs emitted by the n compositions before trimming. ---- */
wfst_status wfst_compose_shortest_path_batch(wfst_ctx* ctx, const wfst_fst* const* acceptors, size_t n,
                                             const wfst_fst* t, const wfst_compose_config* ccfg,
                                             const wfst_shortest_path_config* scfg, wfst_fst** outs,
                                             uint64_t* composed_arcs);

/* ---- host-side mutable VectorFst<TropicalWeight> mirror.  What rustfst-ffi exposes as vec_fst_* /
 * fst_* (rustfst-ffi/src/fst/vector_fst.rs:13-354, src/fst/mod.rs:128-216, src/algorithms/tr_sort.rs:15)
 * for callers that have no Rust VectorFst of their own (the Python mirror, C/C++ programs).  Semantics
 * incl. the property bookkeeping follow rustfst/src/fst_impls/vector_fst/mutable_fst.rs:25-281 and
 * rustfst/src/fst_properties/mutate_properties.rs.  A Rust shim does NOT need these: it flattens its own
 * VectorFst through the trait surface and calls wfst_fst_upload. ---- */
typedef struct wfst_vec_fst wfst_vec_fst;
wfst_status wfst_vec_fst_new(wfst_vec_fst** out);                                   /* vec_fst_new :13 */
wfst_status wfst_vec_fst_destroy(wfst_vec_fst* f);                                  /* fst_destroy mod.rs:376 */
wfst_status wfst_vec_fst_copy(const wfst_vec_fst* f, wfst_vec_fst** out);           /* vec_fst_copy :285 */
wfst_status wfst_vec_fst_add_state(wfst_vec_fst* f, uint32_t* state);               /* vec_fst_add_state :56 */
wfst_status wfst_vec_fst_add_tr(wfst_vec_fst* f, uint32_t state, const wfst_tr* tr); /* vec_fst_add_tr :83 */
wfst_status wfst_vec_fst_set_start(wfst_vec_fst* f, uint32_t state);                /* vec_fst_set_start :26 */
wfst_status wfst_vec_fst_set_final(wfst_vec_fst* f, uint32_t state, float weight);  /* vec_fst_set_final :39 */
wfst_status wfst_vec_fst_del_final_weight(wfst_vec_fst* f, uint32_t state);         /* vec_fst_del_final_weight :101 */
wfst_status wfst_vec_fst_num_states(const wfst_vec_fst* f, uint32_t* n);            /* vec_fst_num_states :248 */
wfst_status wfst_vec_fst_start(const wfst_vec_fst* f, int64_t* start);              /* fst_start mod.rs:128; -1 = None */
/* fst_final_weight mod.rs:143: *is_some = 0 for None */
wfst_status wfst_vec_fst_final_weight(const wfst_vec_fst* f, uint32_t state, float* weight, int* is_some);
wfst_status wfst_vec_fst_num_trs(const wfst_vec_fst* f, uint32_t state, uint64_t* n); /* fst_num_trs mod.rs:162 */
/* fst_get_trs mod.rs:179: copies the state's arcs into out[cap]; *n receives the count */
wfst_status wfst_vec_fst_get_trs(const wfst_vec_fst* f, uint32_t state, wfst_tr* out, uint64_t cap, uint64_t* n);
wfst_status wfst_vec_fst_properties(const wfst_vec_fst* f, uint64_t* props);
wfst_status wfst_vec_fst_tr_sort(wfst_vec_fst* f, int ilabel_cmp);                  /* fst_tr_sort tr_sort.rs:15 */
wfst_status wfst_vec_fst_equals(const wfst_vec_fst* a, const wfst_vec_fst* b, int* equal); /* vec_fst_equals :265 */
/* flatten through the trait surface + upload == the shim's input step */
wfst_status wfst_vec_fst_to_device(wfst_ctx* ctx, const wfst_vec_fst* f, wfst_fst** out);
/* download + rebuild (add_states / set_start / set_trs_unchecked / set_final / set_properties) == the shim's output step */
wfst_status wfst_vec_fst_from_device(const wfst_fst* fst, wfst_vec_fst** out);

/* shortest_path_with_config (shortest_path.rs:76-171) of n FSTs in one call; outs[i] = a new handle each.  With nshortest > 1
 * (unique = false) small inputs — the composed lattices of a decoding batch: BASELINE configs[4] — are searched by ONE
 * launch, one wavefront per input (distances, reverse, the reference's heap search, connect); with nshortest == 1 small
 * inputs (<= 4096 states) are likewise solved by one launch (keys in LDS, the canonical predecessor rule, the walk); larger
 * ones go through the single-FST paths one after the other; with unique = true the distances and arrays of all small inputs
 * come to the host in one launch and the host stages (reversal, determinization, search) run on host threads.  Same
 * results as n calls of wfst_shortest_path. */
wfst_status wfst_shortest_path_batch(wfst_ctx* ctx, const wfst_fst* const* fsts, size_t n, const wfst_shortest_path_config* cfg,
                                     wfst_fst** outs);

/* wfst_compose_shortest_path_batch with the results as RECORDS (the layout of wfst_fst_pack_paths below) instead of handles:
 * out[i * (4 + 4 * max_arcs) ...] = path i, written straight from the kernel's result buffers — for hosts that read the
 * paths as a table (a decoder taking the output labels, the exchange between GPUs).  A path FST handle costs ~0.5 us of
 * host time to build and as much to destroy: beyond a few hundred acceptors per call that, not the GPU, is the batch's
 * time (bench.py `batch_sweep`).  KO if a path has more than max_arcs arcs. */
wfst_status wfst_compose_shortest_path_batch_packed(wfst_ctx* ctx, const wfst_fst* const* acceptors, size_t n, const wfst_fst* t,
                                                    const wfst_compose_config* compose_cfg,
                                                    const wfst_shortest_path_config* sp_cfg, uint32_t max_arcs, uint32_t* out,
                                                    uint64_t* composed_arcs);

/* Packs n linear path FSTs (outputs of the calls above) into fixed-size records for one all-gather:
 * record i = [n_arcs u32, final-weight bits u32, valid u32, 0] followed by max_arcs 16-byte arcs (zero padded),
 * i.e. (4 + 4*max_arcs) u32 words.  KO if a path has more than max_arcs arcs or is not linear. */
wfst_status wfst_fst_pack_paths(const wfst_fst* const* paths, size_t n, uint32_t max_arcs, uint32_t* out);

/* ---- several GPUs: one process or thread per GPU, acceptor i on GPU i mod G, T uploaded on every GPU, no
 * collective during compute; the finished results are all-gathered over RCCL / xGMI.  The reference is single-process
 * (no distributed code to mirror): these are the calls a Rust host adds next to compose / shortest_path (INTEGRATION.md,
 * "8 GPUs from Rust").  librccl is opened on first use; without it these calls return KO and everything else works. ---- */
typedef struct wfst_comm wfst_comm;
#define WFST_COMM_ID_BYTES 128
/* rank 0: a fresh rendezvous id (ncclGetUniqueId), to be handed to every rank by whatever channel the host has */
wfst_status wfst_comm_unique_id(uint8_t* id /* [WFST_COMM_ID_BYTES] */);
/* every rank, collectively (ncclCommInitRank): a communicator bound to ctx's GPU with a stream and pinned staging of its own */
wfst_status wfst_comm_create(wfst_ctx* ctx, const uint8_t* id, uint32_t rank, uint32_t world, wfst_comm** out);
/* The same communicator over a HOST transport: `fn` all-gathers `bytes` bytes per rank between host buffers (recv holds
 * world * bytes, rank-major) and returns 0 — an MPI_Allgather, a gloo group, a test harness.  No RCCL, no device staging,
 * the exchange completes inside _begin; staging sets, record layout and the ragged gather are the code of the RCCL path. */
typedef int (*wfst_allgather_fn)(void* user, const void* send, void* recv, size_t bytes);
wfst_status wfst_comm_create_host(uint32_t rank, uint32_t world, wfst_allgather_fn fn, void* user, wfst_comm** out);
wfst_status wfst_comm_info(const wfst_comm* comm, uint32_t* rank, uint32_t* world);
wfst_status wfst_comm_destroy(wfst_comm* comm);
/* All-gather of n linear path FSTs per rank as fixed-size records (layout of wfst_fst_pack_paths).  _begin packs into
 * pinned memory and queues H2D, ncclAllGather and D2H on the communicator's stream, then returns (the exchange overlaps
 * with whatever the caller does next: the next decoding step); _end waits and copies world * n records, rank-major, to out.
 * One exchange in flight per communicator. */
wfst_status wfst_gather_paths_begin(wfst_comm* comm, const wfst_fst* const* paths, size_t n, uint32_t max_arcs);
wfst_status wfst_gather_paths_end(wfst_comm* comm, uint32_t* out /* [world * n * (4 + 4 * max_arcs)] */);
/* _begin for records that exist already (the table wfst_compose_shortest_path_batch_packed filled): no handles, no packing */
wfst_status wfst_gather_records_begin(wfst_comm* comm, const uint32_t* records, size_t n, uint32_t max_arcs);
/* Orders the communicator's stream behind everything queued on ctx's stream so far: the next exchange then runs AFTER
 * that work (a step's relaxation sweeps need every compute unit; the all-gather kernel is better off beside the start of
 * the next step than in the middle of this one).  Optional; without it an exchange starts as soon as it is queued. */
wfst_status wfst_comm_order_after(wfst_comm* comm, wfst_ctx* ctx);
/* the same for `bytes` opaque bytes per rank */
wfst_status wfst_comm_allgather_begin(wfst_comm* comm, const void* send, size_t bytes);
wfst_status wfst_comm_allgather_end(wfst_comm* comm, void* recv /* [world * bytes] */);
/* ragged: every rank contributes `bytes` bytes (general FSTs in the OpenFST binary format: n-best trees, look-ahead
 * compositions); sizes[r] = bytes of rank r, *recv = one block holding the payloads back to back in rank order
 * (wfst_bytes_destroy), *total = their sum.  Two exchanges: the sizes, then the payloads padded to the largest. */
wfst_status wfst_comm_allgatherv(wfst_comm* comm, const void* send, size_t bytes, uint64_t* sizes /* [world] */, void** recv,
                                 size_t* total);

/* Tie order of shortest_path (nshortest = 1).  0 (default): the canonical rule — fewest arcs, then the smallest (source
 * state, arc position), the smallest final state; schedule-free, what every GPU path computes.  1: the REFERENCE's choice
 * where it is well defined cheaply, i.e. on ACYCLIC inputs (composed lattices): rustfst relaxes states in the topological
 * order of its depth-first visit (queues/auto_queue.rs:23-99 -> TopOrderQueue, top_sort.rs:12-61, dfs_visit.rs:97-187) and
 * keeps the FIRST arc, in (order of the source, arc position), that attains the final distance (shortest_path.rs:214-232),
 * and the first final state in that order.  The visit runs on the host (it is sequential by definition), distances and
 * the predecessor pass on the GPU.  On CYCLIC inputs (and inputs rustfst relaxes LIFO) its choice among tied optima is a
 * function of its whole relaxation history, which no rule reproduces: there, tie order 1 returns the path only when the
 * optimum is UNIQUE (wfst_stats.tied_choices == 0: no state of the path has a second optimal predecessor, one final state
 * attains the optimum) — then it is provably rustfst's path — and is KO ("ambiguous optimum") otherwise, so that a caller
 * that needs rustfst's structure falls back to rustfst (the convention for unsupported cases).  Tie order 0 never fails
 * and reports the same count in wfst_stats.tied_choices. */
wfst_status wfst_ctx_set_tie_order(wfst_ctx* ctx, int reference_order);

/* Share of the device a RESIDENT relaxation launch of this context may occupy (shortest_path on branching FSTs of <= 2M
 * states: sssp_mbox_resident_kernel keeps one 1024-thread workgroup on a compute unit of its own for every block of states,
 * for the whole wide phase of the solve — DESIGN.md §3.2 / §3.3).  0 (default) = the whole device: blocks of 4096 states, up to
 * 245 workgroups for a 1M-state FST — the fastest solve, and nothing else of any size fits beside it: a kernel that holds
 * more than a dozen compute units when the launch arrives makes it WAIT (a 512-string fused batch holds 64 of them for
 * ~0.3 ms).  1 = half the device: blocks of 8192 states where that brings the workgroup count to at most half the compute
 * units (1M states: 123), one launch per level where it does not — the solve alone is ~1.4 x slower (1M states: 0.32 vs
 * 0.22 ms of kernels) and runs BESIDE a batch of that size instead of behind it — or beside ANOTHER half-device query: the
 * device's resident lease has two units, a whole-device solve takes both, a half-device solve one, so two contexts set to 1
 * answer two queries at the same time (1M states: 4.8 k queries/s against 3.8 k one after the other; a third concurrent
 * query, or a whole-device one, takes one launch per level for that solve).  The request class decides: a server whose
 * shortest_path queries share the GPU with large fused batches, or that answers many queries at once, sets 1 on the contexts
 * that run the queries.
 * Same results either way (the keys are the fixed point whatever the block size).  The reference has no counterpart: its
 * algorithms run on one host thread (shortest_path.rs:173-239). */
wfst_status wfst_ctx_set_resident_share(wfst_ctx* ctx, uint32_t share);

/* ---- measurement hooks (bench.py / tests; not part of the reference surface) ---- */
typedef struct {
  /* relaxation kernel (sssp_relax_*): launches, total device time from HIP events on ctx's stream,
   * and the algorithmic units they processed */
  uint64_t relax_launches;
  double relax_ms;
  uint64_t relax_arcs;    /* arcs relaxed (sum over launches) */
  uint64_t relax_states;  /* frontier states expanded (sum over launches) */
  uint64_t sweeps;        /* relaxation sweeps of the last solve */
  /* compose */
  uint64_t compose_states; /* composed states created (pre-trim), last call */
  uint64_t compose_arcs;   /* composed arcs emitted (pre-trim), last call */
  uint64_t compose_retries; /* arena-overflow retries, cumulative */
  double compose_ms;       /* device time of the last compose / fused-batch kernel */
  uint64_t string_problems; /* problems of the last fused batch that took the string o T kernel (fst1 a linear,
                               epsilon-free acceptor, fst2 without input epsilons) */
  uint64_t relax_kernel;    /* kernel of the last relaxation: 0 sssp_relax_kernel (atomic sweeps), 1 sssp_mbox_kernel
                               (owner-computes mailbox launches: WIDE / COLLECT / NARROW, one level per launch), 2 the
                               same with the WIDE levels inside one sssp_mbox_resident_kernel launch, 3 atomic sweeps with
                               their dense levels as binned owner-computes passes (sssp_bin_expand / _apply_kernel) */
  uint64_t nbest_device_problems; /* inputs of the last wfst_shortest_path_batch (nshortest > 1) searched by the wave kernel
                                     (the others went through the host search) */
  uint64_t resident_aborts; /* resident relaxation launches that gave up waiting for their own workgroups (the solve was
                               then repeated with one launch per level; the context tries resident launches again after a
                               pause that doubles with every abort in a row), cumulative */
  uint64_t tied_choices;    /* last wfst_shortest_path (nshortest = 1): states of the returned path that had more than one optimal
                               predecessor, + 1 when several final states attain the optimum.  0 = the optimum is UNIQUE: the
                               path is what rustfst returns whatever its queue discipline.  > 0 = rustfst may return another path
                               of the same weight.  WFST_TIES_UNKNOWN when the call did not count (first query of an FST, FSTs
                               of < 2^18 arcs in the default tie order, tiny inputs, paths beyond 4096 arcs, acyclic inputs
                               under tie order 1 — exact there anyway) */
} wfst_stats;
#define WFST_TIES_UNKNOWN (~(uint64_t)0)
/* on = 1: every relaxation launch is bracketed by HIP events and followed by a synchronisation (per-launch trace below;
 * never on in timed runs).  on = 2: no per-launch events; the sweeps of a repeated shortest_path query (one pre-queued
 * batch) are timed as ONE chain between two events on the stream: relax_ms = that time, relax_launches = its sweeps
 * (0 when the query was not a single predicted batch).  on = 0: off. */
wfst_status wfst_ctx_set_profiling(wfst_ctx* ctx, int on);
wfst_status wfst_ctx_get_stats(wfst_ctx* ctx, wfst_stats* out);
wfst_status wfst_ctx_reset_stats(wfst_ctx* ctx);
/* per-launch trace of the last profiled relaxation (profiling on): launch k relaxed arcs[k] arcs leaving
 * states[k] frontier states in ms[k] milliseconds.  Copies min(cap, *n) entries; arrays may be NULL to query *n. */
wfst_status wfst_ctx_get_sweep_trace(wfst_ctx* ctx, double* ms, uint64_t* arcs, uint64_t* states, size_t cap, size_t* n);
/* ... and what ran launch k: 0 the atomic sweep, 7 a binned level (the same level as an owner-computes pass: expand +
 * apply kernels, chosen per level on the device), or the mailbox launches' mode (0 WIDE, 1 COLLECT, 2 NARROW). */
wfst_status wfst_ctx_get_sweep_modes(wfst_ctx* ctx, uint32_t* modes, size_t cap, size_t* n);

#ifdef __cplusplus
}
#endif
#endif /* WFST_AMD_H */
