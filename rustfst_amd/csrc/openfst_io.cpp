// openfst_io.cpp — OpenFST binary reader ("vector" and "const", "standard" arcs) and "vector" writer (host side).
// const format (CSR on disk, 16-byte aligned blocks when version == 1):
// rustfst/src/fst_impls/const_fst/serializable_fst.rs:176-237, const_fst/mod.rs:11-14.
// Format: rustfst/src/parsers/bin_fst/fst_header.rs:71-137 (header),
// rustfst/src/fst_impls/vector_fst/serializable_fst.rs:45-168 (body, store()),
// rustfst/src/parsers/bin_fst/utils_parsing.rs:10-44 (start / final / arc),
// rustfst/src/parsers/bin_symt/nom_parser.rs:14-45 (symbol tables; skipped: they never reach the device).
#include <algorithm>
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
  const bool is_const = fst_type == "const";
  if (fst_type != "vector" && !is_const)
    throw Error("Error while parsing binary Fst: fst_type is '" + fst_type + "', expected 'vector' or 'const'");
  if (arc_type != "standard")  // Tr::<TropicalWeight>::tr_type(), tr.rs:68-76
    throw Error("Error while parsing binary VectorFst: arc_type is '" + arc_type + "', expected 'standard'");
  const int32_t version = c.get<int32_t>();
  if (version < (is_const ? 1 : 2)) throw Error("Error while parsing binary Fst: unsupported version");  // :127 / mod.rs:11
  uint32_t flags = c.get<uint32_t>();
  if (flags & ~7u) throw Error("Could not parse Fst Flags");
  uint64_t props = c.get<uint64_t>();
  int64_t start = c.get<int64_t>();
  int64_t num_states = c.get<int64_t>();
  const int64_t num_arcs_hdr = c.get<int64_t>();  // may be 0 in vector files (ignored there, :157); exact in const files
  if (flags & 1u) skip_symt(c);
  if (flags & 2u) skip_symt(c);
  if (num_states < 0 || num_states >= 0x7FFFFFFF) throw Error("Error while parsing binary VectorFst: bad num_states");
  // the header is untrusted: every state costs at least 12 bytes of body in a vector file (final weight + arc count) and
  // 20 in a const file, so a state count the remaining bytes cannot hold is refused BEFORE anything is reserved for it
  {
    const uint64_t per_state = is_const ? 20u : 12u;
    if ((uint64_t)num_states > (len - std::min(c.off, len)) / per_state)
      throw Error("Error while parsing binary Fst: num_states exceeds what the file can hold");
  }
  HostCsr h;
  h.offsets.reserve((size_t)num_states + 1);
  h.finals.reserve((size_t)num_states);
  h.offsets.push_back(0);
  if (is_const) {  // parse_const_fst: const_fst/serializable_fst.rs:198-237
    const bool aligned = version == 1;
    auto align16 = [&]() {
      if (aligned && (c.off % 16) != 0) c.off += 16 - (c.off % 16);
      if (c.off > c.n) throw Error("Error while parsing binary ConstFst");
    };
    if (num_arcs_hdr < 0 || (uint64_t)num_arcs_hdr > len / 16) throw Error("Error while parsing binary ConstFst");
    if (num_states > 0) align16();
    std::vector<uint32_t> pos((size_t)num_states), ntrs((size_t)num_states);
    for (int64_t s = 0; s < num_states; ++s) {
      const float fw = c.get<float>();
      pos[(size_t)s] = (uint32_t)c.get<int32_t>();
      ntrs[(size_t)s] = (uint32_t)c.get<int32_t>();
      (void)c.get<int32_t>();  // niepsilons / noepsilons are recomputed on the device
      (void)c.get<int32_t>();
      h.finals.push_back(props::is_zero(fw) ? INF : fw);
    }
    if (num_arcs_hdr > 0) align16();
    h.arcs.resize((size_t)num_arcs_hdr);
    for (int64_t i = 0; i < num_arcs_hdr; ++i) {
      wfst_tr& tr = h.arcs[(size_t)i];
      tr.ilabel = (uint32_t)c.get<int32_t>();
      tr.olabel = (uint32_t)c.get<int32_t>();
      tr.weight = c.get<float>();
      tr.nextstate = (uint32_t)c.get<int32_t>();
    }
    for (int64_t s = 0; s < num_states; ++s) {  // states must tile the arc array in order (they do in OpenFST files)
      if (pos[(size_t)s] != h.offsets.back() || (uint64_t)pos[(size_t)s] + ntrs[(size_t)s] > (uint64_t)num_arcs_hdr)
        throw Error("Error while parsing binary ConstFst: state arc ranges are not contiguous");
      h.offsets.push_back(pos[(size_t)s] + ntrs[(size_t)s]);
    }
    if (start < -1 || start >= num_states) throw Error("Error while parsing binary ConstFst: start out of range");
    return upload_from_host(ctx, (uint32_t)num_states, start, h.offsets.data(), h.arcs.data(), h.finals.data(), props);
  }
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

// ConstFst::store (const_fst/serializable_fst.rs:41-89): version 2 (unaligned); the CSR goes out as it is
void fst_to_openfst_const_bytes(const wfst_fst* f, std::vector<uint8_t>& out) {
  ensure_host(f);
  const HostCsr& h = f->host;
  if (f->n_arcs > 0x7FFFFFFFull) throw Error("FST too large for the const format (i32 arc positions)");
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
  out.reserve(64 + (size_t)f->n_states * 20 + (size_t)f->n_arcs * 16);
  put_i32(FST_MAGIC);
  put_str("const");
  put_str("standard");
  put_i32(2);  // CONST_FILE_VERSION (const_fst/mod.rs:13)
  uint32_t flags = 0;
  put(&flags, 4);
  uint64_t p = f->props | 0x1ull;  // ConstFst::static_properties() = EXPANDED
  put(&p, 8);
  put_i64(f->start);
  put_i64((int64_t)f->n_states);
  put_i64((int64_t)f->n_arcs);
  for (uint32_t s = 0; s < f->n_states; ++s) {
    const uint32_t b = h.offsets[s], e = h.offsets[s + 1];
    int32_t nie = 0, noe = 0;
    for (uint32_t i = b; i < e; ++i) {
      nie += h.arcs[i].ilabel == WFST_EPS_LABEL;
      noe += h.arcs[i].olabel == WFST_EPS_LABEL;
    }
    put(&h.finals[s], 4);
    put_i32((int32_t)b);
    put_i32((int32_t)(e - b));
    put_i32(nie);
    put_i32(noe);
  }
  if (f->n_arcs) put(h.arcs.data(), (size_t)f->n_arcs * sizeof(wfst_tr));  // {i32,i32,f32,i32} LE == wfst_tr
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
