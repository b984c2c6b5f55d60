// openfst_io.cpp — OpenFST binary "vector"/"standard" reader and writer (host side).
// Format: rustfst/src/parsers/bin_fst/fst_header.rs:71-137 (header),
// rustfst/src/fst_impls/vector_fst/serializable_fst.rs:45-168 (body, store()),
// rustfst/src/parsers/bin_fst/utils_parsing.rs:10-44 (start / final / arc),
// rustfst/src/parsers/bin_symt/nom_parser.rs:14-45 (symbol tables; skipped: they never reach the device).
#include "common.h"
#include "fst_props.h"

namespace wfst {

namespace {
constexpr int32_t FST_MAGIC = 2125659606;   // fst_header.rs:18
constexpr int32_t SYMT_MAGIC = 2125658996;  // bin_symt/nom_parser.rs:14

struct Cursor {
  const uint8_t* p;
  size_t n, off = 0;
  template <class T>
  T get() {
    if (off + sizeof(T) > n) throw Error("Error while parsing binary VectorFst: unexpected end of data");
    T v;
    std::memcpy(&v, p + off, sizeof(T));
    off += sizeof(T);
    return v;
  }
  std::string str() {  // OpenFstString: i32 length + bytes
    int32_t len = get<int32_t>();
    if (len < 0 || off + (size_t)len > n) throw Error("Error while parsing binary VectorFst: bad string");
    std::string s((const char*)p + off, (size_t)len);
    off += (size_t)len;
    return s;
  }
};

void skip_symt(Cursor& c) {
  if (c.get<int32_t>() != SYMT_MAGIC) throw Error("Error while parsing symbolTable from binary VectorFst");
  c.str();
  c.get<int64_t>();
  int64_t num = c.get<int64_t>();
  for (int64_t i = 0; i < num; ++i) {
    c.str();
    c.get<int64_t>();
  }
}
}  // namespace

wfst_fst* fst_from_openfst_bytes(wfst_ctx* ctx, const uint8_t* data, size_t len) {
  Cursor c{data, len};
  if (c.get<int32_t>() != FST_MAGIC) throw Error("Error while parsing binary VectorFst: bad magic number");
  std::string fst_type = c.str(), arc_type = c.str();
  if (fst_type != "vector")
    throw Error("Error while parsing binary VectorFst: fst_type is '" + fst_type + "', expected 'vector'");
  if (arc_type != "standard")  // Tr::<TropicalWeight>::tr_type(), tr.rs:68-76
    throw Error("Error while parsing binary VectorFst: arc_type is '" + arc_type + "', expected 'standard'");
  if (c.get<int32_t>() < 2) throw Error("Error while parsing binary VectorFst: version < 2");  // :127
  uint32_t flags = c.get<uint32_t>();
  if (flags & ~7u) throw Error("Could not parse Fst Flags");
  uint64_t props = c.get<uint64_t>();
  int64_t start = c.get<int64_t>();
  int64_t num_states = c.get<int64_t>();
  (void)c.get<int64_t>();  // num_arcs: may be 0 in vector files; the reference ignores it (:157)
  if (flags & 1u) skip_symt(c);
  if (flags & 2u) skip_symt(c);
  if (num_states < 0 || num_states >= 0x7FFFFFFF) throw Error("Error while parsing binary VectorFst: bad num_states");
  HostCsr h;
  h.offsets.reserve((size_t)num_states + 1);
  h.finals.reserve((size_t)num_states);
  h.offsets.push_back(0);
  for (int64_t s = 0; s < num_states; ++s) {
    float fw = c.get<float>();
    int64_t ntrs = c.get<int64_t>();
    if (ntrs < 0 || (uint64_t)ntrs > (len - c.off) / 16) throw Error("Error while parsing binary VectorFst: bad arc count");
    // parse_final_weight (utils_parsing.rs:18-26): None iff weight == zero, i.e. +inf
    h.finals.push_back(props::is_zero(fw) ? INF : fw);
    for (int64_t i = 0; i < ntrs; ++i) {
      wfst_tr tr;
      tr.ilabel = (uint32_t)c.get<int32_t>();
      tr.olabel = (uint32_t)c.get<int32_t>();
      tr.weight = c.get<float>();
      tr.nextstate = (uint32_t)c.get<int32_t>();
      h.arcs.push_back(tr);
    }
    if (h.arcs.size() > 0xFFFFFFFFull) throw Error("FST too large: more than 2^32 arcs");
    h.offsets.push_back((uint32_t)h.arcs.size());
  }
  if (start < -1 || start >= num_states) throw Error("Error while parsing binary VectorFst: start out of range");
  return upload_from_host(ctx, (uint32_t)num_states, start, h.offsets.data(), h.arcs.data(), h.finals.data(), props);
}

void fst_to_openfst_bytes(const wfst_fst* f, std::vector<uint8_t>& out) {
  ensure_host(f);
  const HostCsr& h = f->host;
  auto put = [&](const void* p, size_t n) {
    const uint8_t* b = (const uint8_t*)p;
    out.insert(out.end(), b, b + n);
  };
  auto put_i32 = [&](int32_t v) { put(&v, 4); };
  auto put_i64 = [&](int64_t v) { put(&v, 8); };
  auto put_str = [&](const char* s) {
    put_i32((int32_t)std::strlen(s));
    put(s, std::strlen(s));
  };
  out.reserve(66 + (size_t)f->n_states * 12 + (size_t)f->n_arcs * 16);
  put_i32(FST_MAGIC);
  put_str("vector");
  put_str("standard");
  put_i32(2);  // version
  uint32_t flags = 0;
  put(&flags, 4);
  uint64_t p = f->props | props::STATIC_BITS;  // serializable_fst.rs:65-66
  put(&p, 8);
  put_i64(f->start);
  put_i64((int64_t)f->n_states);
  put_i64((int64_t)f->n_arcs);
  for (uint32_t s = 0; s < f->n_states; ++s) {
    put(&h.finals[s], 4);
    uint32_t b = h.offsets[s], e = h.offsets[s + 1];
    put_i64((int64_t)(e - b));
    if (e > b) put(&h.arcs[b], (size_t)(e - b) * sizeof(wfst_tr));  // {i32,i32,f32,i32} LE == wfst_tr
  }
}

}  // namespace wfst
