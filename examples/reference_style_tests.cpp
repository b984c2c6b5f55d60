// reference_style_tests.cpp — the reference's own known-answer tests for this path, written against wfst.hpp so that they
// read like the originals:
//   K1  rustfst-python/tests/algorithms/test_compose.py:13-81          test_compose_fst
//   K2  rustfst-python/tests/algorithms/test_shortest_path.py:5-51     test_shortest_path
//   K3  rustfst/src/algorithms/compose/compose_static.rs:282-289       doctest of compose (fst![1,2 => 2,3] o fst![2,3 => 3,4])
//   K7  rustfst-python/tests/algorithms/test_project.py:5-97           test_project_input / test_project_output
//   K8  rustfst-python/tests/algorithms/test_connect.py:4-55           test_connect
//   K11 rustfst-python/tests/algorithms/test_rm_epsilon.py:4-54        test_rm_epsilon
//   the look-ahead recipe of rustfst-cli/src/cmds/compose.rs:77-181 on the K1 operands: same language as compose
//   error behaviour of compose on unsorted operands (compose_fst_op.rs:169-197) and of missing states (mutable_fst.rs)
//
//   g++ -std=c++17 -I include examples/reference_style_tests.cpp -L rustfst_amd/lib -lwfst_amd -Wl,-rpath,$PWD/rustfst_amd/lib -o ref_tests
#include <cstdio>
#include <cstring>

#include "wfst.hpp"

using namespace wfst_amd;

static int failures = 0;
#define ASSERT(cond)                                                      \
  do {                                                                    \
    if (!(cond)) {                                                        \
      std::printf("ASSERT FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      ++failures;                                                         \
    }                                                                     \
  } while (0)

static Tr tr(Label il, Label ol, float w, StateId ns) { return Tr{il, ol, w, ns}; }

static void test_compose_fst() {  // K1
  VectorFst fst1;
  StateId s1 = fst1.add_state(), s2 = fst1.add_state(), s3 = fst1.add_state();
  fst1.set_start(s1);
  fst1.set_final(s2);
  fst1.set_final(s3);
  fst1.add_tr(s1, tr(1, 2, 1.0f, s2));
  fst1.add_tr(s1, tr(1, 4, 2.0f, s3));
  fst1.add_tr(s2, tr(3, 5, 2.0f, s2));

  VectorFst fst2;
  s1 = fst2.add_state(), s2 = fst2.add_state(), s3 = fst2.add_state();
  fst2.set_start(s1);
  fst2.set_final(s3);
  fst2.add_tr(s1, tr(2, 6, 1.0f, s2));
  fst2.add_tr(s2, tr(5, 7, 2.5f, s3));
  fst2.add_tr(s3, tr(5, 8, 1.5f, s3));
  fst2.add_tr(s1, tr(4, 9, 3.0f, s3));

  VectorFst expected_fst;
  s1 = expected_fst.add_state(), s2 = expected_fst.add_state(), s3 = expected_fst.add_state();
  StateId s4 = expected_fst.add_state();
  expected_fst.set_start(s1);
  expected_fst.set_final(s3);
  expected_fst.set_final(s4);
  expected_fst.add_tr(s1, tr(1, 6, 2.0f, s2));
  expected_fst.add_tr(s1, tr(1, 9, 5.0f, s3));
  expected_fst.add_tr(s2, tr(3, 7, 4.5f, s4));
  expected_fst.add_tr(s4, tr(3, 8, 3.5f, s4));

  VectorFst fst3 = compose(fst1, fst2);
  ASSERT(fst3 == expected_fst);
  for (ComposeFilterEnum f : {ComposeFilterEnum::SequenceFilter, ComposeFilterEnum::AltSequenceFilter,
                              ComposeFilterEnum::MatchFilter, ComposeFilterEnum::NoMatchFilter,
                              ComposeFilterEnum::TrivialFilter, ComposeFilterEnum::NullFilter})
    ASSERT(compose_with_config(fst1, fst2, ComposeConfig{f, true}) == expected_fst);  // no epsilons: all filters agree
}

static void test_shortest_path() {  // K2
  VectorFst fst1;
  StateId s1 = fst1.add_state(), s2 = fst1.add_state(), s3 = fst1.add_state(), s4 = fst1.add_state();
  fst1.set_start(s1);
  fst1.set_final(s4, 2.0f);
  fst1.add_tr(s1, tr(1, 1, 3.0f, s2));
  fst1.add_tr(s2, tr(2, 2, 2.0f, s2));
  fst1.add_tr(s2, tr(3, 3, 4.0f, s4));
  fst1.add_tr(s1, tr(4, 4, 5.0f, s3));
  fst1.add_tr(s3, tr(5, 5, 4.0f, s4));

  VectorFst expected_fst;
  s1 = expected_fst.add_state(), s2 = expected_fst.add_state(), s3 = expected_fst.add_state();
  expected_fst.set_start(s3);
  expected_fst.set_final(s1, 2.0f);
  expected_fst.add_tr(s3, tr(1, 1, 3.0f, s2));
  expected_fst.add_tr(s2, tr(3, 3, 4.0f, s1));

  const ShortestPathConfig config = ShortestPathConfig{}.with_nshortest(1).with_unique(true);  // ShortestPathConfig(1, True)
  VectorFst shortest = shortest_path_with_config(fst1, config);
  ASSERT(shortest == expected_fst);
  ASSERT(shortest_path(fst1) == expected_fst);

  // two best paths (weights 9 and 11): a tree rooted at a fresh start state, one branch per path (shortest_path.rs:409-518)
  VectorFst two = shortest_path_with_config(fst1, ShortestPathConfig{}.with_nshortest(2));
  ASSERT(two.start().has_value() && two.num_trs(*two.start()) == 2);
}

static void test_compose_doctest() {  // K3: transducer(labels 1,2 -> 2,3) o transducer(2,3 -> 3,4) == transducer(1,2 -> 3,4)
  auto transducer = [](std::vector<Label> in, std::vector<Label> out) {  // utils::transducer (labels_to_fst.rs:40-109), one()
    VectorFst f;
    StateId cur = f.add_state();
    f.set_start(cur);
    for (size_t i = 0; i < in.size(); ++i) {
      StateId nxt = f.add_state();
      f.add_tr(cur, Tr{in[i], out[i], 0.0f, nxt});
      cur = nxt;
    }
    f.set_final(cur);
    return f;
  };
  ASSERT(compose(transducer({1, 2}, {2, 3}), transducer({2, 3}, {3, 4})) == transducer({1, 2}, {3, 4}));
}

static void test_errors() {
  VectorFst a;  // olabels 7 then 2 out of state 0: not O_LABEL_SORTED
  a.add_states(2);
  a.set_start(0);
  a.set_final(1);
  a.add_tr(0, tr(1, 7, 0.0f, 1));
  a.add_tr(0, tr(1, 2, 0.0f, 1));
  VectorFst b;  // ilabels 7 then 2: not I_LABEL_SORTED
  b.add_states(2);
  b.set_start(0);
  b.set_final(1);
  b.add_tr(0, tr(7, 1, 0.0f, 1));
  b.add_tr(0, tr(2, 1, 0.0f, 1));
  bool threw = false;
  try {
    compose(a, b);
  } catch (const Error& e) {
    threw = std::strstr(e.what(), "(sort?)") != nullptr;  // compose_fst_op.rs:194
  }
  ASSERT(threw);
  tr_sort(a, OLabelCompare{});
  tr_sort(b, ILabelCompare{});
  VectorFst ab = compose(a, b);
  ASSERT(ab.num_states() == 2 && ab.num_trs(0) == 2);  // 0 --1:1--> 1 twice over (labels 2 and 7 both match)
  threw = false;
  try {
    VectorFst c;
    c.set_start(3);
  } catch (const Error& e) {
    threw = std::strstr(e.what(), "doesn't exist") != nullptr;
  }
  ASSERT(threw);
  ASSERT(shortest_path(VectorFst()).num_states() == 0);  // no start state -> empty result (shortest_path.rs:185-187)
}

static VectorFst project_input_fst() {  // test_project.py:5-27
  VectorFst f;
  const StateId s1 = f.add_state(), s2 = f.add_state(), s3 = f.add_state();
  f.set_start(s1);
  f.set_final(s3);
  f.add_tr(s1, tr(1, 2, 1.0f, s2));
  f.add_tr(s1, tr(3, 4, 2.0f, s2));
  f.add_tr(s2, tr(4, 5, 3.0f, s3));
  return f;
}
static void test_project() {  // K7
  for (int output = 0; output < 2; ++output) {
    VectorFst expected;
    const StateId s1 = expected.add_state(), s2 = expected.add_state(), s3 = expected.add_state();
    expected.set_start(s1);
    expected.set_final(s3);
    expected.add_tr(s1, output ? tr(2, 2, 1.0f, s2) : tr(1, 1, 1.0f, s2));
    expected.add_tr(s1, output ? tr(4, 4, 2.0f, s2) : tr(3, 3, 2.0f, s2));
    expected.add_tr(s2, output ? tr(5, 5, 3.0f, s3) : tr(4, 4, 3.0f, s3));
    VectorFst f = project_input_fst();
    project(f, output ? ProjectType::ProjectOutput : ProjectType::ProjectInput);
    ASSERT(f == expected);
  }
}

static void test_connect() {  // K8
  VectorFst f;
  for (int i = 0; i < 5; ++i) f.add_state();
  f.set_start(0);
  f.set_final(1, 0.0f);
  f.add_tr(4, tr(1, 2, 1.0f, 0));
  f.add_tr(0, tr(3, 4, 2.0f, 1));
  f.add_tr(1, tr(4, 5, 3.0f, 2));
  f.add_tr(2, tr(4, 6, 4.0f, 3));
  f.add_tr(2, tr(7, 8, 5.0f, 0));
  VectorFst expected;
  for (int i = 0; i < 3; ++i) expected.add_state();
  expected.set_start(0);
  expected.set_final(1, 0.0f);
  expected.add_tr(0, tr(3, 4, 2.0f, 1));
  expected.add_tr(1, tr(4, 5, 3.0f, 2));
  expected.add_tr(2, tr(7, 8, 5.0f, 0));
  connect(f);
  ASSERT(f == expected);
}

static void test_rm_epsilon() {  // K11
  VectorFst f;
  for (int i = 0; i < 4; ++i) f.add_state();
  f.set_start(0);
  f.set_final(3, 1.0f);
  f.add_tr(0, tr(0, 0, 1.0f, 1));
  f.add_tr(1, tr(1, 0, 2.0f, 2));
  f.add_tr(1, tr(0, 2, 3.0f, 2));
  f.add_tr(1, tr(0, 0, 4.0f, 2));
  f.add_tr(2, tr(0, 0, 5.0f, 2));
  f.add_tr(2, tr(0, 0, 5.0f, 3));
  VectorFst expected;
  expected.add_state();
  expected.add_state();
  expected.set_start(0);
  expected.set_final(0, 11.0f);
  expected.set_final(1, 6.0f);
  expected.add_tr(0, tr(0, 2, 4.0f, 1));
  expected.add_tr(0, tr(1, 0, 3.0f, 1));
  rm_epsilon(f);
  ASSERT(f == expected);
}

static void test_lookahead_recipe() {
  // fst![1,2 => 2,3] o fst![2,3 => 3,4] through the look-ahead configuration: one path, input 1 2, output 3 4, weight one
  VectorFst a, b;
  for (int i = 0; i < 3; ++i) { a.add_state(); b.add_state(); }
  a.set_start(0); a.set_final(2);
  b.set_start(0); b.set_final(2);
  a.add_tr(0, tr(1, 2, 0.0f, 1)); a.add_tr(1, tr(2, 3, 0.0f, 2));
  b.add_tr(0, tr(2, 3, 0.0f, 1)); b.add_tr(1, tr(3, 4, 0.0f, 2));
  tr_sort(a, OLabelCompare{});
  tr_sort(b, ILabelCompare{});
  const VectorFst c = compose_lookahead(a, b);
  const VectorFst p = shortest_path(c);
  ASSERT(p.num_states() == 3);
  std::vector<Label> il, ol;
  StateId s = *p.start();
  while (p.num_trs(s)) {
    const Tr t = p.get_trs(s)[0];
    il.push_back(t.ilabel); ol.push_back(t.olabel);
    s = t.nextstate;
  }
  ASSERT((il == std::vector<Label>{1, 2}) && (ol == std::vector<Label>{3, 4}));
}

int main() {
  test_connect();
  test_rm_epsilon();
  test_project();
  test_lookahead_recipe();
  test_compose_fst();
  test_shortest_path();
  test_compose_doctest();
  test_errors();
  std::printf(failures ? "%d assertion(s) FAILED\n" : "all reference-style tests passed\n", failures);
  return failures != 0;
}
