/* decode_batch.c — the drop-in boundary used from plain C: upload a transducer once, compose a batch of linear
 * acceptors against it and take the shortest path of each, all through include/wfst.h.
 *
 *   gcc -std=c99 -I include examples/decode_batch.c -L rustfst_amd/lib -lwfst_amd -Wl,-rpath,$PWD/rustfst_amd/lib -o decode_batch
 *
 * What a rustfst caller writes as   for a in acceptors { shortest_path(compose(a, &t)) }   (compose_static.rs:293-303,
 * shortest_path.rs:76-116) becomes one wfst_compose_shortest_path_batch call.  Exit status 0 = every path has the
 * expected weight. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "wfst.h"

#define CHECK(call)                                              \
  do {                                                           \
    if ((call) != WFST_OK) {                                     \
      char* msg = NULL;                                          \
      wfst_last_error(&msg);                                     \
      fprintf(stderr, "%s failed: %s\n", #call, msg ? msg : "?"); \
      wfst_string_destroy(msg);                                  \
      return 1;                                                  \
    }                                                            \
  } while (0)

int main(void) {
  wfst_ctx* ctx = NULL;
  CHECK(wfst_ctx_create(0, &ctx));

  /* T: a 3-state ring that rewrites label l to l+10 with weight 0.5 per step; every state final (weight 0) */
  enum { NT = 3 };
  uint32_t t_off[NT + 1] = {0, 2, 4, 6};
  wfst_tr t_arcs[6] = {{1, 11, 0.5f, 1}, {2, 12, 0.5f, 1}, {1, 11, 0.5f, 2}, {2, 12, 0.5f, 2}, {1, 11, 0.5f, 0}, {2, 12, 0.5f, 0}};
  float t_fin[NT] = {0.0f, 0.0f, 0.0f};
  const uint64_t I_SORTED = 0x10000000ull, O_SORTED = 0x40000000ull; /* fst_properties/properties.rs:52-59 */
  wfst_fst* t = NULL;
  CHECK(wfst_fst_upload(ctx, NT, 0, t_off, t_arcs, t_fin, I_SORTED, &t));

  /* two linear acceptors: "1 2 1" and "2 2" */
  enum { NA = 2 };
  uint32_t labels0[3] = {1, 2, 1}, labels1[2] = {2, 2};
  const uint32_t* labels[NA] = {labels0, labels1};
  uint32_t len[NA] = {3, 2};
  wfst_fst* acc[NA];
  for (int i = 0; i < NA; ++i) {
    uint32_t n = len[i] + 1, off[8];
    wfst_tr arcs[8];
    float fin[8];
    for (uint32_t s = 0; s < n; ++s) {
      off[s] = s;
      fin[s] = INFINITY; /* +inf = not final (the on-disk sentinel, vector_fst/serializable_fst.rs:78-80) */
      if (s < len[i]) arcs[s] = (wfst_tr){labels[i][s], labels[i][s], 0.0f, s + 1};
    }
    off[n] = len[i];
    off[n - 1] = len[i];
    fin[n - 1] = 0.0f;
    CHECK(wfst_fst_upload(ctx, n, 0, off, arcs, fin, I_SORTED | O_SORTED | 0x10000ull /* ACCEPTOR */, &acc[i]));
  }

  wfst_fst* outs[NA] = {NULL, NULL};
  uint64_t composed_arcs = 0;
  CHECK(wfst_compose_shortest_path_batch(ctx, (const wfst_fst* const*)acc, NA, t, NULL, NULL, outs, &composed_arcs));

  int bad = 0;
  for (int i = 0; i < NA; ++i) {
    uint32_t n = 0;
    uint64_t e = 0, props = 0;
    int64_t start = 0;
    CHECK(wfst_fst_info(outs[i], &n, &e, &start, &props));
    uint32_t off[8];
    wfst_tr arcs[8];
    float fin[8];
    CHECK(wfst_fst_download(outs[i], off, arcs, fin));
    float total = fin[0]; /* the path FST is numbered backwards: state 0 is final (shortest_path.rs:251-276) */
    for (uint64_t k = 0; k < e; ++k) total += arcs[k].weight;
    printf("acceptor %d: path of %llu arcs, weight %.3f, output labels:", i, (unsigned long long)e, total);
    for (int64_t s = start; s > 0; --s) printf(" %u", arcs[off[s]].olabel);
    printf("\n");
    if (e != len[i] || fabsf(total - 0.5f * (float)len[i]) > 1e-5f) bad = 1;
    CHECK(wfst_fst_destroy(outs[i]));
    CHECK(wfst_fst_destroy(acc[i]));
  }
  CHECK(wfst_fst_destroy(t));
  CHECK(wfst_ctx_destroy(ctx));
  return bad;
}
