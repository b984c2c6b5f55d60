#include <cstdio>
#include <vector>
#include <cstdint>
#include <cmath>
// Test program (tests/test_host.py::test_path_props_from_fact_union): linear_path_props_from_facts == linear_path_props on
// every sequence of up to six arcs over the fact combinations one arc can have.  Built with g++ -I<repo>/rustfst_amd/csrc.
#include "fst_props.h"
using namespace wfst::props;
static wfst_tr arc_of(uint32_t f, uint32_t state) {  // facts bits 1,2,4,8,16 -> an arc of `state`
  wfst_tr a{};
  a.ilabel = (f & 2u) ? 0u : 3u;
  a.olabel = (f & 4u) ? 0u : ((f & 1u) ? 5u : a.ilabel);
  a.weight = (f & 8u) ? 0.75f : ((f & 1u) ? 0.0005f : 0.0f);
  a.nextstate = (f & 16u) ? state - 1 : state + 1;
  return a;
}
static bool consistent(uint32_t f) {  // il==0 && ol==0 -> il == ol: fact 1 must be clear; il==0 xor ol==0 -> il != ol
  const bool ie = f & 2u, oe = f & 4u, ne = f & 1u;
  if (ie && oe) return !ne;
  if (ie != oe) return ne;
  return true;
}
int main() {
  long long n = 0, bad = 0;
  std::vector<uint32_t> codes;
  for (uint32_t f = 16; f < 32; ++f) if (consistent(f)) codes.push_back(f);
  const float finals[3] = {0.0f, 0.5f, 1e-5f};
  for (int len = 0; len <= 6; ++len) {
    std::vector<int> idx(len, 0);
    for (;;) {
      std::vector<wfst_tr> arcs(len);
      uint32_t uni = 0;
      for (int k = 0; k < len; ++k) { arcs[k] = arc_of(codes[idx[k]], (uint32_t)k + 1); uni |= path_arc_facts(arcs[k].ilabel, arcs[k].olabel, arcs[k].weight); }
      for (float fw : finals) {
        const uint64_t a = linear_path_props(true, (uint32_t)len, fw, arcs.data());
        const uint64_t b = linear_path_props_from_facts(true, (uint32_t)len, fw, uni);
        ++n; if (a != b) { if (bad < 5) printf("mismatch len %d uni %u: %llx vs %llx\n", len, uni, (unsigned long long)a, (unsigned long long)b); ++bad; }
      }
      int k = len - 1;
      while (k >= 0 && ++idx[k] == (int)codes.size()) idx[k--] = 0;
      if (k < 0) break;
    }
  }
  printf("%lld cases, %lld mismatches (%zu consistent fact codes)\n", n, bad, codes.size());
  return bad != 0;
}
