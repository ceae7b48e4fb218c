// Sanitizer harness of csrc/qasr_blob_check.cpp (built by tests/test_blob_check.py with g++ -fsanitize=address,undefined;
// CPU only).  Reads one VALID blob, then
//  (1) mutates every structural field of the header, of every table record (sampled) in turn with a set of hostile values:
//      the checker must answer QASR_ERR_BLOB for those the field list marks as structural, and never trip the sanitizers;
//  (2) applies `n_random` seeded random corruptions (byte flips / 32-bit stores in the tables, truncations, extensions):
//      the checker must answer QASR_OK or QASR_ERR_BLOB without a sanitizer report, and whenever it answers QASR_OK an
//      independent walk dereferences the first and last byte of every array / record the engine would touch - in an
//      exact-size heap copy, so ASan sees any escape.
// Prints one JSON line with the counts; exit code 0 unless a structural mutation was accepted.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "qasr.h"

static volatile unsigned char g_sink;
static inline uint64_t rup(uint64_t x, uint64_t m) { return (x + m - 1) / m * m; }

// what qasr_engine.hip dereferences for an accepted blob (host tables + device arrays, here all in the heap copy)
static void touch(const unsigned char* p, size_t n, uint64_t off, uint64_t bytes) {
  if (!off || !bytes) return;
  qasr_blob_header h;
  memcpy(&h, p, sizeof h);
  g_sink = p[h.data_off + off];
  g_sink = p[h.data_off + off + bytes - 1];
  (void)n;
}
static void walk(const unsigned char* p, size_t n) {
  qasr_blob_header h;
  memcpy(&h, p, sizeof h);
  std::vector<qasr_tensor_desc> t(h.n_tensors);
  memcpy(t.data(), p + h.tensors_off, sizeof(qasr_tensor_desc) * h.n_tensors);
  std::vector<qasr_domain_desc> d(h.n_domains);
  memcpy(d.data(), p + h.domains_off, sizeof(qasr_domain_desc) * h.n_domains);
  for (uint32_t i = 0; i < h.n_ops; ++i) {
    qasr_op_desc op;
    memcpy(&op, p + h.ops_off + (size_t)i * sizeof op, sizeof op);
    g_sink = (unsigned char)t.at(op.in).channels;
    const uint64_t cp = rup(op.cout, 128), cinp = rup(op.cin, 128), kp = rup(op.kernel, 4);
    const bool w6 = op.flags & QASR_F_W6PACK;
    auto wb = [&](uint64_t e) { return w6 ? e / 4 * 3 : e; };
    if (op.kind == QASR_OP_DW) {
      touch(p, n, op.w_off, wb((uint64_t)op.cout * kp));
      if (!(op.flags & QASR_F_RESADD)) touch(p, n, op.m_off, (uint64_t)op.cout * (kp + 32));
    } else if (op.kind == QASR_OP_PW || op.kind == QASR_OP_DENSE) {
      touch(p, n, op.w_off, wb(cp * cinp * (op.kind == QASR_OP_DENSE ? op.kernel : 1)));
      if (op.flags & QASR_F_RESADD) touch(p, n, op.m_off, cp * 8);
    }
    touch(p, n, op.bias_off, cp * 4);
    touch(p, n, op.sb_off, cp * 4);
    for (int j = 0; j < QASR_MAX_OUTS; ++j) {
      if (op.outs[j].tensor < 0) continue;
      g_sink = (unsigned char)d.at(t.at(op.outs[j].tensor).domain).stride;
      if (op.outs[j].mode == 1) touch(p, n, op.outs[j].m_off, cp * 8);
    }
    for (uint32_t k = 0; k < op.n_panes; ++k) {
      const qasr_pane& s = op.panes[k];
      g_sink = (unsigned char)t.at(s.in).channels;
      touch(p, n, s.w_off, wb(cp * rup(s.cin, 128)));
      touch(p, n, s.bias_off, cp * 4);
      touch(p, n, s.m_off, cp * 8);
      touch(p, n, s.sb_off, cp * 4);
    }
  }
}

static int check(const std::vector<unsigned char>& v) {
  // exact-size heap copy at an odd address offset: ASan red zones on both sides, and no alignment the checker could rely on
  unsigned char* raw = (unsigned char*)malloc(v.size() + 1);
  unsigned char* q = raw + 1;
  memcpy(q, v.data(), v.size());
  char why[200];
  int rc = qasr_blob_check(q, v.size(), why, sizeof why);
  if (rc == QASR_OK) {
    unsigned char* al = (unsigned char*)malloc(v.size());
    memcpy(al, v.data(), v.size());
    walk(al, v.size());
    free(al);
  }
  free(raw);
  return rc;
}

struct Rng {
  uint64_t s;
  uint32_t next() {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(s >> 33);
  }
};

struct Field { size_t off, size; const char* name; };
#define F(T, m) {offsetof(T, m), sizeof(((T*)0)->m), #T "." #m}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<unsigned char> good;
  unsigned char buf[65536];
  size_t got;
  while ((got = fread(buf, 1, sizeof buf, f)) > 0) good.insert(good.end(), buf, buf + got);
  fclose(f);
  const uint64_t seed = strtoull(argv[2], nullptr, 10);
  const int n_random = atoi(argv[3]);
  if (check(good) != QASR_OK) {
    printf("{\"error\": \"the unmodified blob is rejected\"}\n");
    return 1;
  }
  qasr_blob_header h;
  memcpy(&h, good.data(), sizeof h);

  // ---- (1) structural fields, each mutated in turn
  const Field hdr[] = {F(qasr_blob_header, magic), F(qasr_blob_header, version), F(qasr_blob_header, n_tensors), F(qasr_blob_header, n_ops),
                       F(qasr_blob_header, n_domains), F(qasr_blob_header, reserved), F(qasr_blob_header, tensors_off),
                       F(qasr_blob_header, ops_off), F(qasr_blob_header, domains_off), F(qasr_blob_header, data_off),
                       F(qasr_blob_header, total_bytes)};
  const Field ten[] = {F(qasr_tensor_desc, channels), F(qasr_tensor_desc, dtype), F(qasr_tensor_desc, domain), F(qasr_tensor_desc, producer)};
  const Field opf[] = {F(qasr_op_desc, kind), F(qasr_op_desc, in), F(qasr_op_desc, cin), F(qasr_op_desc, cout), F(qasr_op_desc, kernel),
                       F(qasr_op_desc, n_panes), F(qasr_op_desc, w_off), F(qasr_op_desc, bias_off), F(qasr_op_desc, sb_off),
                       F(qasr_op_desc, outs[0].tensor), F(qasr_op_desc, outs[0].m_off), F(qasr_op_desc, outs[0].mode),
                       F(qasr_op_desc, panes[0].in), F(qasr_op_desc, panes[0].cin), F(qasr_op_desc, panes[0].w_off),
                       F(qasr_op_desc, panes[0].bias_off), F(qasr_op_desc, panes[0].m_off)};
  // hostile values: far out of range in every interpretation (index, count, offset, size)
  const uint64_t hostile[] = {0xffffffffffffffffull, 0x7fffffffull, 0x80000000ull, 0xfffffff0ull, (uint64_t)good.size(),
                              (uint64_t)good.size() - 1, 0x0000000100000000ull, 0x40000000ull};
  long n_struct = 0, n_struct_rejected = 0, n_struct_accepted_same = 0;
  std::vector<const char*> accepted;
  const bool verbose = getenv("BLOB_FUZZ_VERBOSE") != nullptr;
  auto mutate = [&](size_t base, const Field& fl, bool require_reject) {
    for (uint64_t hv : hostile) {
      std::vector<unsigned char> v = good;
      uint64_t old = 0;
      memcpy(&old, v.data() + base + fl.off, fl.size);
      uint64_t nv = hv;
      if (fl.size == 4) nv &= 0xffffffffull;
      if (nv == old || nv < 4096) continue;                 // a small value may be another VALID index / enum: not hostile
      memcpy(v.data() + base + fl.off, &nv, fl.size);
      ++n_struct;
      int rc = check(v);
      if (rc == QASR_ERR_BLOB) ++n_struct_rejected;
      else if (require_reject) {
        accepted.push_back(fl.name);
        if (verbose) fprintf(stderr, "accepted: %s at record base %zu: %llx -> %llx\n", fl.name, base, (unsigned long long)old, (unsigned long long)nv);
      }
      else ++n_struct_accepted_same;
    }
  };
  for (const Field& fl : hdr) mutate(0, fl, true);
  for (uint32_t i = 0; i < h.n_tensors; i += (h.n_tensors > 64 ? h.n_tensors / 32 : 1))
    for (const Field& fl : ten) mutate(h.tensors_off + (size_t)i * sizeof(qasr_tensor_desc), fl, true);
  for (uint32_t i = 0; i < h.n_ops; i += (h.n_ops > 64 ? h.n_ops / 32 : 1)) {
    qasr_op_desc op;
    memcpy(&op, good.data() + h.ops_off + (size_t)i * sizeof op, sizeof op);
    for (const Field& fl : opf) {
      // a field the op does not use (absent array = 0, unused pane / output slot) need not be rejected: an offset there is
      // either refused as stray or never dereferenced; the walk above proves the second case
      bool used = true;
      const bool conv = op.kind == QASR_OP_DW || op.kind == QASR_OP_PW || op.kind == QASR_OP_DENSE;
      if (!strncmp(fl.name + 13, "panes", 5)) used = op.n_panes > 0;
      if (!strcmp(fl.name + 13, "w_off") || !strcmp(fl.name + 13, "bias_off")) used = conv;
      if (!strcmp(fl.name + 13, "kernel")) used = conv;       // hostile taps on a conv: the weight extent no longer fits
      if (!strcmp(fl.name + 13, "sb_off")) used = op.sb_off != 0;
      if (!strcmp(fl.name + 13, "outs[0].m_off")) used = op.outs[0].tensor >= 0 && op.outs[0].mode == 1;
      if (!strcmp(fl.name + 13, "outs[0].mode") || !strcmp(fl.name + 13, "outs[0].tensor")) used = op.outs[0].tensor >= 0;
      if (!strcmp(fl.name + 13, "panes[0].bias_off")) used = used && op.panes[0].bias_off != 0;
      mutate(h.ops_off + (size_t)i * sizeof op, fl, used);
    }
  }

  // ---- (2) seeded random corruptions
  Rng r{seed * 2654435761ull + 12345};
  long n_ok = 0, n_rej = 0;
  const size_t table_end = h.data_off;
  for (int it = 0; it < n_random; ++it) {
    std::vector<unsigned char> v = good;
    const uint32_t kind = r.next() % 8;
    if (kind == 0) {                                         // truncate
      v.resize(r.next() % v.size());
    } else if (kind == 1) {                                  // extend with noise
      size_t extra = 1 + r.next() % 4096;
      for (size_t k = 0; k < extra; ++k) v.push_back((unsigned char)r.next());
    } else if (kind == 2) {                                  // truncate AND patch total_bytes so the size test passes
      size_t nn = sizeof(qasr_blob_header) + r.next() % (v.size() - sizeof(qasr_blob_header));
      v.resize(nn);
      uint64_t tb = nn;
      memcpy(v.data() + offsetof(qasr_blob_header, total_bytes), &tb, 8);
    } else {                                                 // 1..8 stores into header / tables
      const int cnt = 1 + r.next() % 8;
      for (int k = 0; k < cnt; ++k) {
        size_t at = r.next() % (table_end - 4);
        uint32_t val;
        switch (r.next() % 4) {
          case 0: val = r.next(); break;
          case 1: val = 0xffffffffu; break;
          case 2: val = r.next() % 4096; break;
          default: { memcpy(&val, v.data() + at, 4); val ^= 1u << (r.next() % 32); }
        }
        if (r.next() & 1) at &= ~(size_t)3;
        memcpy(v.data() + at, &val, 4);
      }
    }
    int rc = v.size() ? check(v) : QASR_ERR_BLOB;
    if (rc == QASR_OK) ++n_ok;
    else if (rc == QASR_ERR_BLOB) ++n_rej;
    else {
      printf("{\"error\": \"unexpected status %d\"}\n", rc);
      return 1;
    }
  }
  printf("{\"structural\": %ld, \"structural_rejected\": %ld, \"unused_field_accepted\": %ld, \"structural_accepted\": %zu, "
         "\"first_accepted\": \"%s\", \"random\": %d, \"random_ok\": %ld, \"random_rejected\": %ld}\n",
         n_struct, n_struct_rejected, n_struct_accepted_same, accepted.size(), accepted.empty() ? "" : accepted[0], n_random, n_ok, n_rej);
  return accepted.empty() ? 0 : 1;
}
