// facade_test.cpp -- exercises the C++ facade (reference class names over the C-ABI) the way the
// reference's own tests do: ValidateFle of fle-test.cc:172-200, ValidateDict of
// dict-test.cc:32-62, plus the predicate / scanner paths the reference never tests, against a
// row-at-a-time model.  Needs a GPU (everything runs through libips_hip.so).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <set>
#include <vector>

#include "../exec/hdfs-parquet-table-writer.h"
#include "../exprs/scalar-fn-call.h"

using namespace impala;

static int g_fail = 0, g_checks = 0;
#define CHECK(cond)                                                          \
  do {                                                                       \
    ++g_checks;                                                              \
    if (!(cond)) {                                                           \
      if (++g_fail <= 20) fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
    }                                                                        \
  } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// ---- fle-test.cc:172-200 -------------------------------------------------------------------
static void ValidateFle(const std::vector<int>& values, int bit_width, int expected_len) {
  const int len = 64 * 1024;
  std::vector<uint8_t> buffer(len);
  FleEncoder encoder(buffer.data(), len, bit_width);
  for (size_t i = 0; i < values.size(); ++i) CHECK(encoder.Put((uint64_t)values[i]));
  int encoded_len = encoder.Flush();
  if (expected_len != -1) CHECK(encoded_len == expected_len);
  FleDecoder decoder(buffer.data(), encoded_len, bit_width);
  for (size_t i = 0; i < values.size(); ++i) {
    uint64_t val = 0;
    bool result = decoder.Get(&val);
    CHECK(result);
    CHECK((uint64_t)values[i] == val);
  }
}

static void TestFleSpecificSequences() {
  std::vector<int> values(100);
  for (int i = 0; i < 50; ++i) values[i] = 0;
  for (int i = 50; i < 100; ++i) values[i] = 1;
  ValidateFle(values, 1, 16);
  ValidateFle(values, 2, 32);
  for (int width = 2; width <= 32; ++width) ValidateFle(values, width, -1);
  for (int i = 0; i < 100; ++i) values[i] = i % 2;
  ValidateFle(values, 1, 16);
  for (int width = 2; width <= 32; ++width) ValidateFle(values, width, -1);
}

static void TestFleValues() {
  for (int width = 1; width <= 32; ++width) {
    const uint64_t mod = 1ull << width;
    for (int variant = 0; variant < 4; ++variant) {
      std::vector<int> values;
      int n = variant == 0 ? 1 : 1024;
      for (int v = 0; v < n; ++v)
        values.push_back(variant == 2 ? 0 : variant == 3 ? 1 : (int)((uint64_t)v % mod));
      ValidateFle(values, width, -1);
    }
    std::vector<int> values;
    for (int v = 0; v < 1024; ++v) values.push_back((int)(rnd() % mod & 0x7fffffff));
    ValidateFle(values, width, -1);
  }
}

// ---- predicates + cursor (no reference test exists: row model) --------------------------------
static void TestFlePredicates() {
  for (int width : {1, 3, 8, 11, 16, 21, 32}) {
    const uint64_t mod = 1ull << width;
    const int n = 5000;
    std::vector<uint32_t> vals(n);
    std::vector<uint8_t> buffer((size_t)ips_fle_encoded_bytes(n, width));
    FleEncoder enc(buffer.data(), (int)buffer.size(), width);
    for (int i = 0; i < n; ++i) { vals[i] = (uint32_t)(rnd() % mod); enc.Put(vals[i]); }
    int len = enc.Flush();
    FleDecoder dec(buffer.data(), len, width);
    uint64_t c = vals[17];
    // reference-shaped batches of 1024 rows with Skip in between concatenate to the full answer
    for (int op = 0; op < 5; ++op) {
      FleDecoder d2(buffer.data(), len, width);
      int64_t done = 0;
      while (done < n) {
        int64_t batch = std::min<int64_t>(1024, n - done);
        SkipBitset bs;
        switch (op) {
          case 0: d2.Eq(batch, bs, c); break;
          case 1: d2.Lt(batch, bs, c); break;
          case 2: d2.Le(batch, bs, c); break;
          case 3: d2.Gt(batch, bs, c); break;
          default: d2.Ge(batch, bs, c); break;
        }
        CHECK((int64_t)bs.size() == batch);
        for (int64_t i = 0; i < batch; ++i) {
          uint64_t x = vals[done + i];
          bool e = op == 0 ? x == c : op == 1 ? x < c : op == 2 ? x <= c : op == 3 ? x > c : x >= c;
          CHECK(bs[(size_t)i] == e);
        }
        done += batch;
        if (done < n) CHECK(d2.Skip((int)batch));
      }
    }
    // mid-block start + non-advancing + Get(v, skip)
    uint32_t v;
    CHECK(dec.Get(&v, 70) && v == vals[70]);
    CHECK(dec.Get(&v, 0) && v == vals[71]);
    SkipBitset bs;
    std::vector<uint64_t> in_list = {vals[100], vals[200], mod - 1};
    dec.In(300, bs, in_list);
    for (int i = 0; i < 300; ++i) {
      uint64_t x = vals[72 + i];
      CHECK(bs[(size_t)i] == (x == in_list[0] || x == in_list[1] || x == in_list[2]));
    }
    CHECK(dec.Get(&v) && v == vals[72]);
    // end of data
    FleDecoder d3(buffer.data(), len, width);
    CHECK(d3.Skip(n - 1));
    CHECK(d3.Get(&v) && v == vals[n - 1]);
    CHECK(!d3.Skip(64 * 3));
  }
}

// ---- dict-test.cc:32-62, 118-147 ------------------------------------------------------------
template <typename T>
static void ValidateDict(const std::vector<T>& values) {
  std::set<T> values_set(values.begin(), values.end());
  DictEncoder<T> encoder(nullptr, -1);
  for (const T& i : values) encoder.Put(i);
  CHECK(encoder.num_entries() == (int)values_set.size());
  std::vector<uint8_t> dict_buffer((size_t)encoder.dict_encoded_size() + 8);
  encoder.WriteDict(dict_buffer.data());
  std::vector<uint8_t> data_buffer(1 << 20);
  int data_len = encoder.WriteData(data_buffer.data(), (int)data_buffer.size());
  CHECK(data_len > 0);
  encoder.ClearIndices();
  DictDecoder<T> decoder(dict_buffer.data(), encoder.dict_encoded_size(), -1);
  decoder.SetData(data_buffer.data(), data_len);
  for (const T& i : values) {
    T j;
    CHECK(decoder.GetValue(&j));
    CHECK(i == j);
  }
  // the bulk GPU encoder produces the same two pages byte for byte
  const int slot = ips_plain_stride(IpsTypeOf<T>::value);
  std::vector<uint8_t> slots(values.size() * (size_t)slot);
  for (size_t i = 0; i < values.size(); ++i) ParquetPlainEncoder::Encode(slots.data() + i * slot, -1, values[i]);
  ips::DeviceBuffer d_slots;
  CHECK(d_slots.upload(slots.data(), slots.size()));
  std::vector<uint8_t> gpu_dict, gpu_data;
  CHECK(DictEncoder<T>::EncodeColumn(d_slots.get(), (int64_t)values.size(), &gpu_dict, &gpu_data));
  CHECK((int)gpu_dict.size() == encoder.dict_encoded_size());
  CHECK(memcmp(gpu_dict.data(), dict_buffer.data(), gpu_dict.size()) == 0);
  CHECK((int)gpu_data.size() == data_len);
  CHECK(memcmp(gpu_data.data(), data_buffer.data(), (size_t)data_len) == 0);
}

template <typename T>
static void TestNumbers(int max_value, int repeat) {
  std::vector<T> values;
  for (int val = 0; val < max_value; ++val)
    for (int i = 0; i < repeat; ++i) values.push_back((T)val);
  ValidateDict(values);
}

template <typename T>
static void TestNumbersAll() {
  TestNumbers<T>(100, 1);
  TestNumbers<T>(1, 100);
  TestNumbers<T>(1, 1);
  TestNumbers<T>(1, 2);
}

template <typename T>
static void TestDictPredicates() {
  const int n = 6000;
  std::vector<T> pool;
  for (int i = 0; i < 300; ++i) pool.push_back((T)((int)(rnd() % 20000) - 10000) + (T)(std::is_floating_point<T>::value ? 0.25 : 0));
  std::vector<T> vals(n);
  DictEncoder<T> encoder;
  for (int i = 0; i < n; ++i) { vals[i] = i < 300 ? pool[i] : pool[rnd() % 300]; encoder.Put(vals[i]); }
  std::vector<uint8_t> dict_buffer((size_t)encoder.dict_encoded_size() + 8), data_buffer(1 << 20);
  encoder.WriteDict(dict_buffer.data());
  int data_len = encoder.WriteData(data_buffer.data(), (int)data_buffer.size());
  DictDecoder<T> decoder(dict_buffer.data(), encoder.dict_encoded_size(), -1);
  decoder.SetData(data_buffer.data(), data_len);
  T lo = *std::min_element(pool.begin(), pool.end()), hi = *std::max_element(pool.begin(), pool.end());
  std::vector<T> lits = {(T)(lo - 1), lo, pool[5], (T)(pool[5] + (std::is_floating_point<T>::value ? 0.125 : 0)), hi, (T)(hi + 1)};
  for (T lit : lits) {
    for (int op = 0; op < 5; ++op) {
      SkipBitset bs;
      switch (op) {
        case 0: decoder.Eq(n, bs, lit); break;
        case 1: decoder.Lt(n, bs, lit); break;
        case 2: decoder.Le(n, bs, lit); break;
        case 3: decoder.Gt(n, bs, lit); break;
        default: decoder.Ge(n, bs, lit); break;
      }
      CHECK((int)bs.size() == n);
      for (int i = 0; i < n; ++i) {
        T x = vals[i];
        bool e = op == 0 ? x == lit : op == 1 ? x < lit : op == 2 ? x <= lit : op == 3 ? x > lit : x >= lit;
        CHECK(bs[(size_t)i] == e);
      }
    }
  }
  std::vector<T> in = {pool[1], (T)(hi + 5), pool[250]};
  SkipBitset bs;
  decoder.In(n, bs, in);
  for (int i = 0; i < n; ++i) CHECK(bs[(size_t)i] == (vals[i] == in[0] || vals[i] == in[2]));
}

// ---- scanner: batches == fused program == row model; nullable; PLAIN; late materialisation ----
static void TestScanner() {
  const int n = 10000;
  std::vector<int32_t> c0(n);
  std::vector<int64_t> c1(n);
  std::vector<int32_t> c2(n), c3(n);
  std::vector<bool> c3_null(n);
  DictEncoder<int32_t> e0, e3;
  DictEncoder<int64_t> e1;
  std::vector<uint32_t> defs;
  for (int i = 0; i < n; ++i) {
    c0[i] = (int32_t)(rnd() % 2526);
    c1[i] = (int64_t)(rnd() % 11) * 1000000007ll;
    c2[i] = (int32_t)(rnd() % 2000) - 1000;
    c3_null[i] = (rnd() % 5) == 0;
    c3[i] = (int32_t)(rnd() % 50);
    e0.Put(c0[i]); e1.Put(c1[i]);
    if (!c3_null[i]) e3.Put(c3[i]);
    defs.push_back(c3_null[i] ? 0u : 1u);
  }
  auto pages = [](auto& enc, std::vector<uint8_t>& dict, std::vector<uint8_t>& data) {
    dict.resize((size_t)enc.dict_encoded_size() + 8);
    enc.WriteDict(dict.data());
    data.resize(1 << 20);
    int len = enc.WriteData(data.data(), (int)data.size());
    data.resize((size_t)len);
    return enc.dict_encoded_size();
  };
  std::vector<uint8_t> d0, p0, d1, p1, d3, p3;
  int dl0 = pages(e0, d0, p0), dl1 = pages(e1, d1, p1), dl3 = pages(e3, d3, p3);
  // OPTIONAL column page: [int32 n_def_bytes][FLE def levels bw=1][uint8 w][codes]
  std::vector<uint8_t> defbuf((size_t)ips_fle_encoded_bytes(n, 1));
  FleEncoder defenc(defbuf.data(), (int)defbuf.size(), 1);
  for (uint32_t d : defs) defenc.Put(d);
  int32_t n_def_bytes = defenc.Flush();
  std::vector<uint8_t> p3full(4 + (size_t)n_def_bytes + p3.size());
  memcpy(p3full.data(), &n_def_bytes, 4);
  memcpy(p3full.data() + 4, defbuf.data(), (size_t)n_def_bytes);
  memcpy(p3full.data() + 4 + n_def_bytes, p3.data(), p3.size());
  std::vector<uint8_t> plain2((size_t)n * 4);
  memcpy(plain2.data(), c2.data(), plain2.size());

  auto build = [&](HdfsParquetScanner& s, bool with_nullable) {
    s.AddDictionaryColumn<int32_t>(d0.data(), dl0, p0.data(), (int)p0.size(), n);
    s.AddDictionaryColumn<int64_t>(d1.data(), dl1, p1.data(), (int)p1.size(), n);
    s.AddPlainColumn<int32_t>(plain2.data(), n);
    if (with_nullable) s.AddDictionaryColumn<int32_t>(d3.data(), dl3, p3full.data(), (int)p3full.size(), n, 1);
  };
  // conjuncts through the plan-time lowering: c0 >= 300 AND c0 < 1500; c1 IN (..) OR c2 > 500
  auto conjuncts = [&](std::vector<ExprContext*>& ctxs, bool with_nullable) {
    ctxs.push_back(new ExprContext(new AndPredicate(
        new ScalarFnCall("ge", new SlotRef(TYPE_INT, 0), {new Literal(TYPE_INT, (int64_t)300)}),
        new ScalarFnCall("lt", new Cast(TYPE_INT, new SlotRef(TYPE_INT, 0)), {new Literal(TYPE_INT, (int64_t)1500)}))));
    ctxs.push_back(new ExprContext(new OrPredicate(
        new ScalarFnCall("in_set_lookup", new SlotRef(TYPE_BIGINT, 1),
                         {new Literal(TYPE_BIGINT, (int64_t)(3 * 1000000007ll)), new Literal(TYPE_BIGINT, (int64_t)(7 * 1000000007ll)),
                          new Literal(TYPE_BIGINT, (int64_t)12345)}),
        new ScalarFnCall("lt", new SlotRef(TYPE_INT, 2), {new Literal(TYPE_INT, (int64_t)500)}))));
    if (with_nullable)
      ctxs.push_back(new ExprContext(new ScalarFnCall("lt", new SlotRef(TYPE_INT, 3), {new Literal(TYPE_INT, (int64_t)24)})));
  };
  auto expect = [&](int i, bool with_nullable) {
    bool a = c0[i] >= 300 && c0[i] < 1500;
    // PLAIN leaves use the reference operand order (literal OP x): "lt 500" means 500 < x
    bool b = (c1[i] == 3 * 1000000007ll || c1[i] == 7 * 1000000007ll) || (500 < c2[i]);
    bool c = !with_nullable || (!c3_null[i] && c3[i] < 24);
    return a && b && c;
  };

  for (int with_nullable = 0; with_nullable < 2; ++with_nullable) {
    HdfsParquetScanner scanner;
    build(scanner, with_nullable);
    std::vector<ExprContext*> ctxs;
    conjuncts(ctxs, with_nullable);
    std::vector<SimplePredicate*> roots;
    CHECK(CreateSimplePredicates(&scanner, ctxs, &roots));
    for (SimplePredicate* r : roots) scanner.AddSimplePredicate(r);
    // the vectorised loop of AssembleRows (.cc:1101-1182): batches, skip list, late materialisation
    int64_t row = 0;
    int64_t selected = 0;
    while (row < n) {
      SkipBitset bs;
      CHECK(scanner.EvalSimplePredicates(bs));
      const int64_t batch = (int64_t)bs.size();
      CHECK(batch == std::min<int64_t>(1024, n - row));
      for (int64_t i = 0; i < batch; ++i) CHECK(bs[(size_t)i] == expect((int)(row + i), with_nullable));
      std::vector<int> skip_rows;
      int last_skip_rows = 0;
      HdfsParquetScanner::BitsetToSkipList(bs, &skip_rows, &last_skip_rows);
      int64_t r = row;
      for (int skip : skip_rows) {
        r += skip;
        int32_t v0 = 0; int64_t v1 = 0; int32_t v2 = 0;
        CHECK(scanner.ReadValue(0, &v0, skip) && v0 == c0[r]);
        CHECK(scanner.ReadValue(1, &v1, skip) && v1 == c1[r]);
        CHECK(scanner.ReadValue(2, &v2, skip) && v2 == c2[r]);
        if (with_nullable) {
          int32_t v3; bool is_null;
          CHECK(scanner.ReadValue(3, &v3, skip, &is_null) && !is_null && v3 == c3[r]);
        }
        ++r; ++selected;
      }
      if (last_skip_rows) {
        for (int c = 0; c < 3 + with_nullable; ++c) scanner.SkipValue(c, last_skip_rows);
      }
      row += batch;
    }
    CHECK(selected > 0);
    {  // the whole conjunct list over all rows in one ips_eval_program call: REQUIRED, PLAIN and
       // (with_nullable) the OPTIONAL dictionary column through the fused nullable leaf
      HdfsParquetScanner s2;
      build(s2, with_nullable);
      std::vector<ExprContext*> ctxs2;
      conjuncts(ctxs2, with_nullable);
      std::vector<SimplePredicate*> roots2;
      CHECK(CreateSimplePredicates(&s2, ctxs2, &roots2));
      for (SimplePredicate* r : roots2) s2.AddSimplePredicate(r);
      std::vector<uint64_t> words;
      CHECK(s2.EvalSimplePredicatesFused(n, &words));
      for (int i = 0; i < n; ++i) CHECK((bool)((words[(size_t)i >> 6] >> (i & 63)) & 1) == expect(i, with_nullable));
      for (ExprContext* c : ctxs2) delete c;
    }
    for (ExprContext* c : ctxs) delete c;
  }
  // shapes the reference refuses to lower (scalar-fn-call.cc:740-746, 945-962)
  HdfsParquetScanner s;
  ExprContext slot_vs_slot(new ScalarFnCall("eq", new SlotRef(TYPE_INT, 0), {new SlotRef(TYPE_INT, 1)}));
  ExprContext bool_lit(new ScalarFnCall("eq", new SlotRef(TYPE_INT, 0), {new Literal(TYPE_BOOLEAN, (int64_t)1)}));
  ExprContext unknown_fn(new ScalarFnCall("ne", new SlotRef(TYPE_INT, 0), {new Literal(TYPE_INT, (int64_t)1)}));
  CHECK(slot_vs_slot.CreateSimplePredicates(&s) == NULL);
  CHECK(bool_lit.CreateSimplePredicates(&s) == NULL);
  CHECK(unknown_fn.CreateSimplePredicates(&s) == NULL);
}

// A column chunk as several data pages per column, page boundaries differing between the columns
// (dictionary int32: 3000 + 5000 + 2011 rows; dictionary OPTIONAL int32: 4000 + 6011; PLAIN int32:
// 7000 + 3011): the batches stop at every page end of every column (.cc:1839-1850) and the next
// page is picked up by the following call.
static void TestScannerMultiPage() {
  const int n = 10011;
  std::vector<int32_t> c0(n), c1(n), c2(n);
  std::vector<bool> c1_null(n);
  DictEncoder<int32_t> e0, e1;
  for (int i = 0; i < n; ++i) {
    c0[i] = (int32_t)(rnd() % 700);
    c1_null[i] = (rnd() % 7) == 0;
    c1[i] = (int32_t)(rnd() % 90) - 40;
    c2[i] = (int32_t)(rnd() % 4000) - 2000;
    e0.Put(c0[i]);
    if (!c1_null[i]) e1.Put(c1[i]);
  }
  std::vector<uint8_t> d0((size_t)e0.dict_encoded_size() + 8), d1((size_t)e1.dict_encoded_size() + 8);
  e0.WriteDict(d0.data());
  e1.WriteDict(d1.data());
  // one data page = the indices of its rows, written against the finished dictionary
  auto data_page = [&](DictEncoder<int32_t>& enc, const std::vector<int32_t>& col, const std::vector<bool>* nulls,
                       int lo, int hi) {
    enc.ClearIndices();
    std::vector<uint32_t> defs;
    for (int i = lo; i < hi; ++i) {
      const bool is_null = nulls && (*nulls)[(size_t)i];
      if (!is_null) enc.Put(col[(size_t)i]);
      defs.push_back(is_null ? 0u : 1u);
    }
    std::vector<uint8_t> codes(1 << 18);
    const int len = enc.WriteData(codes.data(), (int)codes.size());
    CHECK(len > 0);
    codes.resize((size_t)len);
    if (!nulls) return codes;
    std::vector<uint8_t> defbuf((size_t)ips_fle_encoded_bytes(hi - lo, 1));
    FleEncoder defenc(defbuf.data(), (int)defbuf.size(), 1);
    for (uint32_t d : defs) defenc.Put(d);
    const int32_t n_def_bytes = defenc.Flush();
    std::vector<uint8_t> page(4 + (size_t)n_def_bytes + codes.size());
    memcpy(page.data(), &n_def_bytes, 4);
    memcpy(page.data() + 4, defbuf.data(), (size_t)n_def_bytes);
    memcpy(page.data() + 4 + n_def_bytes, codes.data(), codes.size());
    return page;
  };
  const int b0[] = {0, 3000, 8000, n}, b1[] = {0, 4000, n}, b2[] = {0, 7000, n};
  std::vector<std::vector<uint8_t>> p0, p1;
  for (int k = 0; k < 3; ++k) p0.push_back(data_page(e0, c0, nullptr, b0[k], b0[k + 1]));
  for (int k = 0; k < 2; ++k) p1.push_back(data_page(e1, c1, &c1_null, b1[k], b1[k + 1]));
  std::vector<uint8_t> plain((size_t)n * 4);
  memcpy(plain.data(), c2.data(), plain.size());

  HdfsParquetScanner scanner;
  scanner.AddDictionaryColumn<int32_t>(d0.data(), e0.dict_encoded_size(), p0[0].data(), (int)p0[0].size(), b0[1] - b0[0]);
  scanner.AddDictionaryColumn<int32_t>(d1.data(), e1.dict_encoded_size(), p1[0].data(), (int)p1[0].size(), b1[1] - b1[0], 1);
  scanner.AddPlainColumn<int32_t>(plain.data(), b2[1]);
  for (int k = 1; k < 3; ++k) scanner.AddDataPage(0, p0[(size_t)k].data(), (int)p0[(size_t)k].size(), b0[k + 1] - b0[k]);
  scanner.AddDataPage(1, p1[1].data(), (int)p1[1].size(), b1[2] - b1[1]);
  scanner.AddDataPage(2, plain.data() + (size_t)b2[1] * 4, (n - b2[1]) * 4, n - b2[1]);

  std::vector<ExprContext*> ctxs;
  ctxs.push_back(new ExprContext(new ScalarFnCall("lt", new SlotRef(TYPE_INT, 0), {new Literal(TYPE_INT, (int64_t)350)})));
  ctxs.push_back(new ExprContext(new OrPredicate(
      new ScalarFnCall("ge", new SlotRef(TYPE_INT, 1), {new Literal(TYPE_INT, (int64_t)10)}),
      new ScalarFnCall("gt", new SlotRef(TYPE_INT, 2), {new Literal(TYPE_INT, (int64_t)1500)}))));
  std::vector<SimplePredicate*> roots;
  CHECK(CreateSimplePredicates(&scanner, ctxs, &roots));
  for (SimplePredicate* r : roots) scanner.AddSimplePredicate(r);
  auto expect = [&](int i) {
    // PLAIN leaves use the reference operand order (literal OP x): "gt 1500" means 1500 > x
    return c0[(size_t)i] < 350 && ((!c1_null[(size_t)i] && c1[(size_t)i] >= 10) || 1500 > c2[(size_t)i]);
  };
  // the same pages through the page-list evaluation: every page of the three chunks in one
  // ips_eval_program_chunks call, page ends at 3000 / 8000, 4000 and 7000 rows
  {
    HdfsParquetScanner s2;
    s2.AddDictionaryColumn<int32_t>(d0.data(), e0.dict_encoded_size(), p0[0].data(), (int)p0[0].size(), b0[1] - b0[0]);
    s2.AddDictionaryColumn<int32_t>(d1.data(), e1.dict_encoded_size(), p1[0].data(), (int)p1[0].size(), b1[1] - b1[0], 1);
    s2.AddPlainColumn<int32_t>(plain.data(), b2[1]);
    for (int k = 1; k < 3; ++k) s2.AddDataPage(0, p0[(size_t)k].data(), (int)p0[(size_t)k].size(), b0[k + 1] - b0[k]);
    s2.AddDataPage(1, p1[1].data(), (int)p1[1].size(), b1[2] - b1[1]);
    s2.AddDataPage(2, plain.data() + (size_t)b2[1] * 4, (n - b2[1]) * 4, n - b2[1]);
    std::vector<SimplePredicate*> roots2;
    CHECK(CreateSimplePredicates(&s2, ctxs, &roots2));
    for (SimplePredicate* r : roots2) s2.AddSimplePredicate(r);
    std::vector<uint64_t> words;
    int64_t rows = 0;
    CHECK(s2.EvalSimplePredicatesChunks(&words, &rows));
    CHECK(rows == n && (int64_t)words.size() == (n + 63) / 64);
    int64_t wrong = 0;
    for (int i = 0; i < n; ++i) wrong += (((words[(size_t)i >> 6] >> (i & 63)) & 1) != 0) != expect(i);
    CHECK(wrong == 0);
    CHECK((words.back() >> (n & 63)) == 0);  // bits behind the last row are zero
  }
  // rows left in each column's current page bound every batch
  const int cuts[] = {3000, 4000, 7000, 8000, n};
  int64_t row = 0, selected = 0;
  int batches = 0;
  while (row < n) {
    SkipBitset bs;
    CHECK(scanner.EvalSimplePredicates(bs));
    const int64_t batch = (int64_t)bs.size();
    int64_t to_cut = n - row;
    for (int c : cuts) if (c > row) { to_cut = c - row; break; }
    CHECK(batch == std::min<int64_t>(1024, to_cut));
    if (batch <= 0) break;
    for (int64_t i = 0; i < batch; ++i) CHECK(bs[(size_t)i] == expect((int)(row + i)));
    std::vector<int> skip_rows;
    int last_skip_rows = 0;
    HdfsParquetScanner::BitsetToSkipList(bs, &skip_rows, &last_skip_rows);
    int64_t r = row;
    for (int skip : skip_rows) {
      r += skip;
      int32_t v0 = 0, v1 = 0, v2 = 0;
      bool is_null = false;
      CHECK(scanner.ReadValue(0, &v0, skip) && v0 == c0[(size_t)r]);
      CHECK(scanner.ReadValue(1, &v1, skip, &is_null) && is_null == (bool)c1_null[(size_t)r] && (is_null || v1 == c1[(size_t)r]));
      CHECK(scanner.ReadValue(2, &v2, skip) && v2 == c2[(size_t)r]);
      ++r; ++selected;
    }
    if (last_skip_rows)
      for (int c = 0; c < 3; ++c) scanner.SkipValue(c, last_skip_rows);
    row += batch;
    ++batches;
  }
  CHECK(row == n && selected > 100 && batches == 10);  // 3 + 1 + 3 + 1 + 2 batches between the page cuts
  SkipBitset tail;
  CHECK(!scanner.EvalSimplePredicates(tail));  // no page left in any column
  for (ExprContext* c : ctxs) delete c;
}

// The reference's call pattern -- every leaf once per 1024-row batch (hdfs-parquet-scanner.cc:1838)
// -- must not turn into a device launch, an upload or a download per batch: a BETWEEN (And(Ge, Le),
// simple-predicates.h:145-153) over a 2^20-row page is 2 whole-page evaluations, not 2048.
static void TestCallPatternLaunchCounts() {
  const int n = 1 << 20;
  std::vector<int32_t> c0(n), c1(n), c2(n);
  std::vector<bool> c1_null(n);
  DictEncoder<int32_t> e0, e1;
  for (int i = 0; i < n; ++i) {
    c0[i] = (int32_t)(rnd() % 3000);
    c1_null[i] = (rnd() % 9) == 0;
    c1[i] = (int32_t)(rnd() % 200);
    c2[i] = (int32_t)(rnd() % 100000);
    e0.Put(c0[i]);
    if (!c1_null[i]) e1.Put(c1[i]);
  }
  auto pages = [](DictEncoder<int32_t>& enc, std::vector<uint8_t>& dict, std::vector<uint8_t>& data) {
    dict.resize((size_t)enc.dict_encoded_size() + 8);
    enc.WriteDict(dict.data());
    data.resize((size_t)4 << 20);
    int len = enc.WriteData(data.data(), (int)data.size());
    data.resize((size_t)len);
  };
  std::vector<uint8_t> d0, p0, d1, p1;
  pages(e0, d0, p0);
  pages(e1, d1, p1);
  std::vector<uint8_t> defbuf((size_t)ips_fle_encoded_bytes(n, 1));
  FleEncoder defenc(defbuf.data(), (int)defbuf.size(), 1);
  for (int i = 0; i < n; ++i) defenc.Put(c1_null[i] ? 0u : 1u);
  int32_t n_def_bytes = defenc.Flush();
  std::vector<uint8_t> p1full(4 + (size_t)n_def_bytes + p1.size());
  memcpy(p1full.data(), &n_def_bytes, 4);
  memcpy(p1full.data() + 4, defbuf.data(), (size_t)n_def_bytes);
  memcpy(p1full.data() + 4 + n_def_bytes, p1.data(), p1.size());
  std::vector<uint8_t> plain((size_t)n * 4);
  memcpy(plain.data(), c2.data(), plain.size());

  // which: 0 REQUIRED dictionary column, 1 OPTIONAL dictionary column, 2 PLAIN column
  for (int which = 0; which < 3; ++which) {
    HdfsParquetScanner scanner;
    const int lo = which == 0 ? 500 : which == 1 ? 20 : 60000, hi = which == 0 ? 1500 : which == 1 ? 120 : 1000;
    if (which == 0) scanner.AddDictionaryColumn<int32_t>(d0.data(), e0.dict_encoded_size(), p0.data(), (int)p0.size(), n);
    if (which == 1) scanner.AddDictionaryColumn<int32_t>(d1.data(), e1.dict_encoded_size(), p1full.data(), (int)p1full.size(), n, 1);
    if (which == 2) scanner.AddPlainColumn<int32_t>(plain.data(), n);
    SimplePredicate* between = scanner.Own(new AndOperate(scanner.Own(new GeOperate<int32_t>(0, lo)),
                                                          scanner.Own(new LeOperate<int32_t>(0, hi))));
    scanner.AddSimplePredicate(between);
    const ips::FacadeStats before = ips::stats();
    int64_t row = 0, bad = 0;
    while (row < n) {
      SkipBitset bs;
      CHECK(scanner.EvalSimplePredicates(bs));
      const int64_t batch = (int64_t)bs.size();
      if (batch <= 0) break;
      for (int64_t i = 0; i < batch; ++i) {
        const int64_t r = row + i;
        bool e;
        if (which == 0) e = c0[r] >= lo && c0[r] <= hi;
        else if (which == 1) e = !c1_null[r] && c1[r] >= lo && c1[r] <= hi;
        else e = lo >= c2[r] && hi <= c2[r];  // PLAIN: the reference's operand order, literal OP x (quirk Q1)
        if (bs[(size_t)i] != e) ++bad;
      }
      // the predicates do not advance the decoders (fle-encoding.h:7962-8313): the batch is
      // consumed by the materialisation that follows, here a SkipValue of the whole batch
      CHECK(scanner.SkipValue(0, (int)batch) || row + batch == n);
      row += batch;
    }
    CHECK(row == n && bad == 0);
    const ips::FacadeStats after = ips::stats();
    // Ge + Le (+ def levels == max_def for the OPTIONAL column), each ONCE for the page
    CHECK(after.pred_launches - before.pred_launches <= (which == 1 ? 3 : 2));
    CHECK(after.decode_launches - before.decode_launches <= (which == 1 ? 1 : 0));  // SkipValue walks the def levels: one page decode
    if (which == 2) CHECK(after.page_uploads - before.page_uploads == 1);  // the PLAIN page, once
  }
}

// Truncated / corrupt data pages are refused instead of read out of bounds.
static void TestTruncatedPages() {
  DictEncoder<int32_t> e;
  for (int i = 0; i < 200; ++i) e.Put(i % 17);
  std::vector<uint8_t> dict((size_t)e.dict_encoded_size() + 8), data(4096);
  e.WriteDict(dict.data());
  int len = e.WriteData(data.data(), (int)data.size());
  std::vector<uint8_t> page(4 + 32 + (size_t)len);
  int32_t n_def = 32;
  memcpy(page.data(), &n_def, 4);
  memset(page.data() + 4, 0xFF, 32);
  memcpy(page.data() + 36, data.data(), (size_t)len);
  HdfsParquetScanner s;
  CHECK(s.AddDictionaryColumn<int32_t>(dict.data(), e.dict_encoded_size(), page.data(), (int)page.size(), 200, 1) == 0);
  CHECK(ips::sticky_status() == IPS_OK);
  CHECK(s.AddDictionaryColumn<int32_t>(dict.data(), e.dict_encoded_size(), page.data(), 3, 200, 1) == -1);      // < 4 bytes
  int32_t huge = 1 << 30;
  memcpy(page.data(), &huge, 4);
  CHECK(s.AddDictionaryColumn<int32_t>(dict.data(), e.dict_encoded_size(), page.data(), (int)page.size(), 200, 1) == -1);  // n_def_bytes > left
  int32_t neg = -8;
  memcpy(page.data(), &neg, 4);
  CHECK(s.AddDictionaryColumn<int32_t>(dict.data(), e.dict_encoded_size(), page.data(), (int)page.size(), 200, 1) == -1);  // negative
  memcpy(page.data(), &n_def, 4);
  CHECK(s.AddDictionaryColumn<int32_t>(dict.data(), e.dict_encoded_size(), page.data(), 36, 200, 1) == -1);     // no width byte left
  // a page header that claims more rows than the level / code blocks hold: refused before
  // anything is uploaded -- the fused kernels would read past the device allocations
  CHECK(s.AddDictionaryColumn<int32_t>(dict.data(), e.dict_encoded_size(), page.data(), (int)page.size(), 257, 1) == -1);     // levels cover 256 rows
  CHECK(s.AddDictionaryColumn<int32_t>(dict.data(), e.dict_encoded_size(), page.data(), (int)page.size(), 256, 1) >= 0);      // (whole blocks: 256 row slots exist)
  CHECK(s.AddDictionaryColumn<int32_t>(dict.data(), e.dict_encoded_size(), data.data(), len, 100000, 0) == -1);               // REQUIRED: codes for 200 rows
  CHECK(s.AddDictionaryColumn<int32_t>(dict.data(), e.dict_encoded_size(), data.data(), len, 200, 0) >= 0);
  {
    HdfsParquetScanner lying;
    CHECK(lying.AddDictionaryColumn<int32_t>(dict.data(), e.dict_encoded_size(), data.data(), len, 200, 0) == 0);
    lying.AddSimplePredicate(lying.Own(new LtOperate<int32_t>(0, 5)));
    std::vector<uint64_t> words;
    CHECK(!lying.EvalSimplePredicatesFused(100000, &words));   // more rows than the page holds
    CHECK(lying.EvalSimplePredicatesFused(200, &words));
    lying.AddDataPage(0, data.data(), len, 100000);            // a queued page that lies: the chunk is refused
    int64_t rows = 0;
    CHECK(!lying.EvalSimplePredicatesChunks(&words, &rows));
  }
  page[36] = 77;
  CHECK(s.AddDictionaryColumn<int32_t>(dict.data(), e.dict_encoded_size(), page.data(), (int)page.size(), 200, 1) == -1);  // width 77
  CHECK(ips::sticky_status() == IPS_ERR_INVALID_ARG);
  ips::sticky_status() = IPS_OK;
}

// The page container: column chunks written the way BaseColumnWriter::Flush lays
// them out (hdfs-parquet-table-writer.cc:478-620: thrift PageHeader + dictionary page, then
// PageHeader + [n_def_bytes][levels][width][codes] per data page, GZIP or uncompressed) are walked
// by AddColumnChunk (ReadDataPage, hdfs-parquet-scanner.cc:730-924) and scanned.
static void TestColumnChunkStream() {
  const int n = 30011;
  std::vector<int32_t> c0(n);
  std::vector<int64_t> c1(n);
  std::vector<bool> c1_null(n);
  for (int i = 0; i < n; ++i) {
    c0[i] = (int32_t)(rnd() % 900) - 300;
    c1_null[i] = (rnd() % 6) == 0;
    c1[i] = (int64_t)(rnd() % 300) * 1000003ll;
  }
  for (int codec : {(int)parquet::CompressionCodec::UNCOMPRESSED, (int)parquet::CompressionCodec::GZIP,
                    (int)parquet::CompressionCodec::SNAPPY}) {
    ColumnChunkWriter<int32_t> w0(codec, 0, 7000);   // REQUIRED, 5 data pages
    ColumnChunkWriter<int64_t> w1(codec, 1, 11000);  // OPTIONAL, 3 data pages
    for (int i = 0; i < n; ++i) {
      CHECK(w0.AppendRow(&c0[i]));
      CHECK(w1.AppendRow(c1_null[i] ? nullptr : &c1[i]));
    }
    std::vector<uint8_t> chunk0, chunk1;
    CHECK(w0.Flush(&chunk0) && w1.Flush(&chunk1));
    CHECK(w0.num_values() == n && w1.num_values() == n);
    HdfsParquetScanner scanner;
    std::string err;
    CHECK(scanner.AddColumnChunk<int32_t>(chunk0.data(), (int64_t)chunk0.size(), n, codec, 0, &err) == 0);
    CHECK(scanner.AddColumnChunk<int64_t>(chunk1.data(), (int64_t)chunk1.size(), n, codec, 1, &err) == 1);
    CHECK(err.empty());
    SimplePredicate* p0 = scanner.Own(new AndOperate(scanner.Own(new GeOperate<int32_t>(0, -100)),
                                                     scanner.Own(new LtOperate<int32_t>(0, 350))));
    SimplePredicate* p1 = scanner.Own(new LeOperate<int64_t>(1, 150 * 1000003ll));
    scanner.AddSimplePredicate(p0);
    scanner.AddSimplePredicate(p1);
    auto expect = [&](int i) { return c0[i] >= -100 && c0[i] < 350 && !c1_null[i] && c1[i] <= 150 * 1000003ll; };
    {  // the whole chunks (5 pages of 7000 rows against 3 of 11000) in one page-list evaluation
      std::vector<uint64_t> words;
      int64_t rows = 0;
      CHECK(scanner.EvalSimplePredicatesChunks(&words, &rows));
      CHECK(rows == n);
      int64_t wrong = 0;
      for (int i = 0; i < n; ++i) wrong += (((words[(size_t)i >> 6] >> (i & 63)) & 1) != 0) != expect(i);
      CHECK(wrong == 0);
    }
    int64_t row = 0, bad = 0, selected = 0;
    while (row < n) {
      SkipBitset bs;
      CHECK(scanner.EvalSimplePredicates(bs));
      const int64_t batch = (int64_t)bs.size();
      if (batch <= 0) break;
      for (int64_t i = 0; i < batch; ++i) if (bs[(size_t)i] != expect((int)(row + i))) ++bad;
      std::vector<int> skip_rows;
      int last_skip_rows = 0;
      HdfsParquetScanner::BitsetToSkipList(bs, &skip_rows, &last_skip_rows);
      int64_t r = row;
      for (int skip : skip_rows) {
        r += skip;
        int32_t v0 = 0; int64_t v1 = 0; bool is_null = true;
        CHECK(scanner.ReadValue(0, &v0, skip) && v0 == c0[r]);
        CHECK(scanner.ReadValue(1, &v1, skip, &is_null) && !is_null && v1 == c1[r]);
        ++r; ++selected;
      }
      if (last_skip_rows) { scanner.SkipValue(0, last_skip_rows); scanner.SkipValue(1, last_skip_rows); }
      row += batch;
    }
    CHECK(row == n && bad == 0 && selected > 0);
    CHECK(ips::sticky_status() == IPS_OK);

    // what ReadDataPage refuses
    HdfsParquetScanner s2;
    CHECK(s2.AddColumnChunk<int32_t>(chunk0.data(), (int64_t)chunk0.size(), n, parquet::CompressionCodec::LZO, 0, &err) == -1);
    if (codec != parquet::CompressionCodec::UNCOMPRESSED)   // the chunk's pages under another codec's name
      CHECK(s2.AddColumnChunk<int32_t>(chunk0.data(), (int64_t)chunk0.size(), n,
                                       codec == parquet::CompressionCodec::GZIP ? parquet::CompressionCodec::SNAPPY
                                                                                : parquet::CompressionCodec::GZIP, 0, &err) == -1);
    CHECK(s2.AddColumnChunk<int32_t>(chunk0.data(), (int64_t)chunk0.size() / 2, n, codec, 0, &err) == -1);   // cut mid-chunk
    CHECK(s2.AddColumnChunk<int32_t>(chunk0.data(), (int64_t)chunk0.size(), n + 5, codec, 0, &err) == -1);  // metadata overstates
    CHECK(s2.AddColumnChunk<int64_t>(chunk0.data(), (int64_t)chunk0.size(), n, codec, 0, &err) == -1);      // wrong slot width
    std::vector<uint8_t> twice(chunk0);                                                                      // two dictionary pages
    {
      parquet::PageHeader h;
      uint32_t hl = (uint32_t)chunk0.size();
      CHECK(parquet::DeserializeThriftMsg(chunk0.data(), &hl, true, &h) && h.type == parquet::PageType::DICTIONARY_PAGE);
      twice.insert(twice.begin(), chunk0.begin(), chunk0.begin() + hl + h.compressed_page_size);
    }
    CHECK(s2.AddColumnChunk<int32_t>(twice.data(), (int64_t)twice.size(), n, codec, 0, &err) == -1);
    CHECK(err.find("two dictionary pages") != std::string::npos);
    // levels declared RLE on an OPTIONAL column (quirk Q11): refused, not misread
    {
      std::vector<uint8_t> rle;
      int64_t pos = 0;
      bool first_data = true;
      while (pos < (int64_t)chunk1.size()) {
        parquet::PageHeader h;
        uint32_t hl = (uint32_t)(chunk1.size() - (size_t)pos);
        CHECK(parquet::DeserializeThriftMsg(chunk1.data() + pos, &hl, true, &h));
        if (h.type == parquet::PageType::DATA_PAGE && first_data) {
          h.data_page_header.definition_level_encoding = parquet::Encoding::RLE;
          first_data = false;
        }
        parquet::SerializePageHeader(h, &rle);
        rle.insert(rle.end(), chunk1.begin() + pos + hl, chunk1.begin() + pos + hl + h.compressed_page_size);
        pos += hl + h.compressed_page_size;
      }
      CHECK(s2.AddColumnChunk<int64_t>(rle.data(), (int64_t)rle.size(), n, codec, 1, &err) == -1);
      CHECK(err.find("FLE") != std::string::npos);
    }
    CHECK(ips::sticky_status() == IPS_ERR_INVALID_ARG);
    ips::sticky_status() = IPS_OK;
  }
}

// From the bytes of the column chunks to row-major tuples on the GPU: AddColumnChunk (the page
// container), the conjunct list, every slot's late materialisation and the tuple assembly
// (AssembleRows' vector path, hdfs-parquet-scanner.cc:1101-1182) -- against a row-at-a-time model
// of InitTuple + ReadValue per selected row, NULL indicator bits included.
static void TestAssembleRowsFused() {
  const int n = 9000;  // one page per column: the fused path works on the current pages
  std::vector<int32_t> c0(n), c2(n);
  std::vector<int64_t> c1(n);
  std::vector<bool> c1_null(n);
  ColumnChunkWriter<int32_t> w0(parquet::CompressionCodec::GZIP, 0, 1 << 20);
  ColumnChunkWriter<int64_t> w1(parquet::CompressionCodec::GZIP, 1, 1 << 20);
  for (int i = 0; i < n; ++i) {
    c0[i] = (int32_t)(rnd() % 1000);
    c1_null[i] = (rnd() % 4) == 0;
    c1[i] = (int64_t)(rnd() % 500) * 4000000007ll - 77;
    c2[i] = (int32_t)(rnd() % 100000) - 50000;
    CHECK(w0.AppendRow(&c0[i]));
    CHECK(w1.AppendRow(c1_null[i] ? nullptr : &c1[i]));
  }
  std::vector<uint8_t> chunk0, chunk1, plain((size_t)n * 4);
  CHECK(w0.Flush(&chunk0) && w1.Flush(&chunk1));
  memcpy(plain.data(), c2.data(), plain.size());
  HdfsParquetScanner scanner;
  std::string err;
  CHECK(scanner.AddColumnChunk<int32_t>(chunk0.data(), (int64_t)chunk0.size(), n, parquet::CompressionCodec::GZIP, 0, &err) == 0);
  CHECK(scanner.AddColumnChunk<int64_t>(chunk1.data(), (int64_t)chunk1.size(), n, parquet::CompressionCodec::GZIP, 1, &err) == 1);
  CHECK(scanner.AddPlainColumn<int32_t>(plain.data(), n) == 2);
  scanner.AddSimplePredicate(scanner.Own(new AndOperate(scanner.Own(new GeOperate<int32_t>(0, 100)),
                                                        scanner.Own(new LtOperate<int32_t>(0, 400)))));
  // tuple: [int32 c0 @0][null byte @4][3 bytes of template][int64 c1 @8][int32 c2 @16][pad @20..23]
  const int tuple_size = 24;
  uint8_t tmpl[tuple_size];
  for (int i = 0; i < tuple_size; ++i) tmpl[i] = (uint8_t)(0xA0 + i);
  tmpl[4] = 0x40;  // other NULL-indicator bits of the byte survive
  std::vector<HdfsParquetScanner::SlotDesc> slots = {{0, 0, 0, 0}, {1, 8, 4, 0x02}, {2, 16, 0, 0}};
  for (int with_pred = 1; with_pred >= 0; --with_pred) {
    if (!with_pred) scanner.ClearSimplePredicates();
    std::vector<uint8_t> tuples;
    int64_t n_tuples = -1;
    CHECK(scanner.AssembleRowsFused(n, tuple_size, tmpl, slots, &tuples, &n_tuples));
    int64_t k = 0, bad = 0;
    for (int i = 0; i < n; ++i) {
      if (with_pred && !(c0[i] >= 100 && c0[i] < 400)) continue;
      if (k >= n_tuples) { ++bad; break; }
      uint8_t expect[tuple_size];
      memcpy(expect, tmpl, tuple_size);                 // InitTuple(template_tuple_, tuple)
      memcpy(expect + 0, &c0[i], 4);
      if (c1_null[i]) expect[4] |= 0x02; else memcpy(expect + 8, &c1[i], 8);
      memcpy(expect + 16, &c2[i], 4);
      if (memcmp(expect, tuples.data() + (size_t)k * tuple_size, tuple_size) != 0) ++bad;
      ++k;
    }
    CHECK(k == n_tuples && bad == 0 && n_tuples > 0);
  }
  // only OPTIONAL slots: the batch counts come from the bitmap
  std::vector<HdfsParquetScanner::SlotDesc> only_opt = {{1, 8, 4, 0x02}};
  std::vector<uint8_t> tuples;
  int64_t n_tuples = -1;
  CHECK(scanner.AssembleRowsFused(n, tuple_size, tmpl, only_opt, &tuples, &n_tuples) && n_tuples == n);
  int64_t bad = 0;
  for (int i = 0; i < n; ++i) {
    const uint8_t* t = tuples.data() + (size_t)i * tuple_size;
    int64_t v;
    memcpy(&v, t + 8, 8);
    if (c1_null[i] ? !(t[4] & 0x02) : ((t[4] & 0x02) || v != c1[i])) ++bad;
  }
  CHECK(bad == 0);
  CHECK(ips::sticky_status() == IPS_OK);
}

// ips_eval_program straight through the C-ABI from a plain C++ process: an OR of two conjunctions
// needs a temporary bitmap next to d_bitmap (stream-ordered allocation inside the call).
// InOperate with more literals than a program node or a kernel argument holds (the reference's In()
// takes a vector of any length): the per-batch pattern and the fused one agree with the row model.
static void TestLongInList() {
  const int n = 20011;
  std::vector<int32_t> c0(n);
  DictEncoder<int32_t> e0;
  for (int i = 0; i < n; ++i) { c0[i] = (int32_t)(rnd() % 5000) * 3; e0.Put(c0[i]); }
  std::vector<uint8_t> d0((size_t)e0.dict_encoded_size() + 8), data(1 << 18);
  e0.WriteDict(d0.data());
  const int len = e0.WriteData(data.data(), (int)data.size());
  CHECK(len > 0);
  std::vector<int32_t> lits;
  for (int v = 0; v < 1200; ++v) lits.push_back(v * 7);  // 1200 literals, a third of them multiples of 3
  std::vector<bool> member(15000 * 3, false);
  for (int32_t v : lits) if (v % 3 == 0) member[(size_t)v] = true;
  for (int pass = 0; pass < 2; ++pass) {
    HdfsParquetScanner scanner;
    CHECK(scanner.AddDictionaryColumn<int32_t>(d0.data(), e0.dict_encoded_size(), data.data(), len, n) == 0);
    scanner.AddSimplePredicate(scanner.Own(new InOperate<int32_t>(0, lits)));
    int64_t wrong = 0;
    if (pass == 0) {
      std::vector<uint64_t> words;
      CHECK(scanner.EvalSimplePredicatesFused(n, &words));
      for (int i = 0; i < n; ++i) wrong += (((words[(size_t)i >> 6] >> (i & 63)) & 1) != 0) != (bool)member[(size_t)c0[i]];
    } else {
      int64_t row = 0;
      while (row < n) {
        SkipBitset bs;
        CHECK(scanner.EvalSimplePredicates(bs));
        for (size_t i = 0; i < bs.size(); ++i) wrong += bs[i] != (bool)member[(size_t)c0[(size_t)row + i]];
        row += (int64_t)bs.size();
        if (bs.size() == 0) break;
        scanner.SkipValue(0, (int)bs.size());  // (the predicates do not advance the row cursor, A.3)
      }
      CHECK(row == n);
    }
    if (wrong) fprintf(stderr, "TestLongInList: pass %d: %lld rows wrong\n", pass, (long long)wrong);
    CHECK(wrong == 0);
  }
  CHECK(ips::sticky_status() == IPS_OK);
}

// Tuples straight from column chunks of several pages whose ends differ between the columns: the page
// loop inside the launches, every slot materialised over its own page list, dense slots assembled.
static void TestAssembleRowsChunks() {
  const int n = 50021;
  std::vector<int32_t> c0(n), c2(n);
  std::vector<int64_t> c1(n);
  DictEncoder<int32_t> e0;
  DictEncoder<int64_t> e1;
  for (int i = 0; i < n; ++i) {
    c0[i] = (int32_t)(rnd() % 900) - 450;
    c1[i] = (int64_t)(rnd() % 77) * 1000003ll - 5;
    c2[i] = (int32_t)(rnd() % 4000) - 2000;
    e0.Put(c0[i]);
    e1.Put(c1[i]);
  }
  std::vector<uint8_t> d0((size_t)e0.dict_encoded_size() + 8), d1((size_t)e1.dict_encoded_size() + 8);
  e0.WriteDict(d0.data());
  e1.WriteDict(d1.data());
  auto page32 = [&](int lo, int hi) {
    e0.ClearIndices();
    for (int i = lo; i < hi; ++i) e0.Put(c0[(size_t)i]);
    std::vector<uint8_t> b(1 << 19);
    const int len = e0.WriteData(b.data(), (int)b.size());
    CHECK(len > 0);
    b.resize((size_t)len);
    return b;
  };
  auto page64 = [&](int lo, int hi) {
    e1.ClearIndices();
    for (int i = lo; i < hi; ++i) e1.Put(c1[(size_t)i]);
    std::vector<uint8_t> b(1 << 19);
    const int len = e1.WriteData(b.data(), (int)b.size());
    CHECK(len > 0);
    b.resize((size_t)len);
    return b;
  };
  // an OPTIONAL int32 dictionary column, ~25 % NULL, four pages ([int32 n_def_bytes][FLE levels][width][codes])
  std::vector<int32_t> c3(n);
  std::vector<char> c3_null(n);
  DictEncoder<int32_t> e3;
  for (int i = 0; i < n; ++i) {
    c3_null[i] = (rnd() % 4) == 0;
    c3[i] = (int32_t)(rnd() % 300) * 7 - 1000;
    if (!c3_null[i]) e3.Put(c3[i]);
  }
  std::vector<uint8_t> d3((size_t)e3.dict_encoded_size() + 8);
  e3.WriteDict(d3.data());
  auto page_opt = [&](int lo, int hi) {
    e3.ClearIndices();
    std::vector<uint8_t> defbuf((size_t)ips_fle_encoded_bytes(hi - lo, 1));
    FleEncoder defenc(defbuf.data(), (int)defbuf.size(), 1);
    for (int i = lo; i < hi; ++i) {
      defenc.Put(c3_null[(size_t)i] ? 0u : 1u);
      if (!c3_null[(size_t)i]) e3.Put(c3[(size_t)i]);
    }
    const int32_t n_def_bytes = defenc.Flush();
    std::vector<uint8_t> codes(1 << 19);
    const int len = e3.WriteData(codes.data(), (int)codes.size());
    CHECK(len > 0);
    std::vector<uint8_t> b(4 + (size_t)n_def_bytes + (size_t)len);
    memcpy(b.data(), &n_def_bytes, 4);
    memcpy(b.data() + 4, defbuf.data(), (size_t)n_def_bytes);
    memcpy(b.data() + 4 + n_def_bytes, codes.data(), (size_t)len);
    return b;
  };
  const int b0[] = {0, 7001, 30000, n}, b1[] = {0, 20011, n}, b2[] = {0, 33, 4100, 41000, n}, b3[] = {0, 12345, 12346, 40000, n};
  std::vector<std::vector<uint8_t>> p3;
  for (int k = 0; k < 4; ++k) p3.push_back(page_opt(b3[k], b3[k + 1]));
  std::vector<std::vector<uint8_t>> p0, p1;
  for (int k = 0; k < 3; ++k) p0.push_back(page32(b0[k], b0[k + 1]));
  for (int k = 0; k < 2; ++k) p1.push_back(page64(b1[k], b1[k + 1]));
  std::vector<uint8_t> plain((size_t)n * 4);
  memcpy(plain.data(), c2.data(), plain.size());
  HdfsParquetScanner s;
  s.AddDictionaryColumn<int32_t>(d0.data(), e0.dict_encoded_size(), p0[0].data(), (int)p0[0].size(), b0[1] - b0[0]);
  s.AddDictionaryColumn<int64_t>(d1.data(), e1.dict_encoded_size(), p1[0].data(), (int)p1[0].size(), b1[1] - b1[0]);
  s.AddPlainColumn<int32_t>(plain.data(), b2[1]);
  for (int k = 1; k < 3; ++k) s.AddDataPage(0, p0[(size_t)k].data(), (int)p0[(size_t)k].size(), b0[k + 1] - b0[k]);
  s.AddDataPage(1, p1[1].data(), (int)p1[1].size(), b1[2] - b1[1]);
  for (int k = 1; k < 4; ++k) s.AddDataPage(2, plain.data() + (size_t)b2[k] * 4, (b2[k + 1] - b2[k]) * 4, b2[k + 1] - b2[k]);
  CHECK(s.AddDictionaryColumn<int32_t>(d3.data(), e3.dict_encoded_size(), p3[0].data(), (int)p3[0].size(), b3[1] - b3[0], 1) == 3);
  for (int k = 1; k < 4; ++k) s.AddDataPage(3, p3[(size_t)k].data(), (int)p3[(size_t)k].size(), b3[k + 1] - b3[k]);
  s.AddSimplePredicate(s.Own(new AndOperate(s.Own(new GeOperate<int32_t>(0, -100)), s.Own(new LtOperate<int32_t>(0, 120)))));
  s.AddSimplePredicate(s.Own(new LeOperate<int64_t>(1, 30 * 1000003ll)));
  // tuple: [int64 c1 @0][int32 c0 @8][int32 c2 @12][int32 c3 (OPTIONAL) @16][NULL indicators @20, bit 0x04 = c3][3 bytes of template] = 24 bytes
  const int ts = 24;
  uint8_t tmpl[ts];
  for (int i = 0; i < ts; ++i) tmpl[i] = (uint8_t)(0xA0 + i);
  tmpl[20] = 0x01;  // (another slot's indicator bit, already set in the template: it must survive)
  std::vector<HdfsParquetScanner::SlotDesc> slots = {{1, 0, 0, 0}, {0, 8, 0, 0}, {2, 12, 0, 0}, {3, 16, 20, 0x04}};
  std::vector<uint8_t> tuples;
  int64_t nt = 0;
  CHECK(s.AssembleRowsChunks(ts, tmpl, slots, &tuples, &nt));
  int64_t exp = 0, wrong = 0, nulls = 0;
  for (int i = 0; i < n; ++i) {
    if (!(c0[(size_t)i] >= -100 && c0[(size_t)i] < 120 && c1[(size_t)i] <= 30 * 1000003ll)) continue;
    if (exp < nt) {
      const uint8_t* t = tuples.data() + (size_t)exp * ts;
      int64_t v1; int32_t v0, v2, v3;
      memcpy(&v1, t, 8); memcpy(&v0, t + 8, 4); memcpy(&v2, t + 12, 4); memcpy(&v3, t + 16, 4);
      wrong += v1 != c1[(size_t)i] || v0 != c0[(size_t)i] || v2 != c2[(size_t)i] || memcmp(t + 21, tmpl + 21, 3) != 0;
      if (c3_null[(size_t)i]) {  // the slot keeps the template's bytes, the indicator bit is set
        wrong += memcmp(t + 16, tmpl + 16, 4) != 0 || t[20] != (0x01 | 0x04);
        ++nulls;
      } else {
        wrong += v3 != c3[(size_t)i] || t[20] != 0x01;
      }
    }
    ++exp;
  }
  CHECK(nt == exp && exp > 1000 && wrong == 0 && nulls > 100);
  CHECK(ips::sticky_status() == IPS_OK);
}

static void TestProgramWithTemporaryBitmap() {
  const int n = 70001;
  std::vector<uint32_t> a(n), b(n);
  for (int i = 0; i < n; ++i) { a[i] = (uint32_t)(rnd() % 4096); b[i] = (uint32_t)(rnd() % 64); }
  ips::DeviceBuffer da, db, ea((size_t)ips_fle_encoded_bytes(n, 12)), eb((size_t)ips_fle_encoded_bytes(n, 6));
  CHECK(da.upload(a.data(), (size_t)n * 4) && db.upload(b.data(), (size_t)n * 4));
  CHECK(ips_fle_encode(da.get(), 4, n, 12, ea.get(), nullptr) == IPS_OK);
  CHECK(ips_fle_encode(db.get(), 4, n, 6, eb.get(), nullptr) == IPS_OK);
  ips_column cols[2] = {{IPS_COL_FLE, 12, 0, 0, ea.get(), nullptr, 0, 0, 0}, {IPS_COL_FLE, 6, 0, 0, eb.get(), nullptr, 0, 0, 0}};
  auto leaf = [](int col, ips_op op, uint64_t c) {
    ips_node nd;
    memset(&nd, 0, sizeof(nd));
    nd.kind = IPS_NODE_LEAF; nd.column = col; nd.op = op; nd.n_consts = 1; nd.consts[0] = c;
    return nd;
  };
  auto inner = [](ips_node_kind k) { ips_node nd; memset(&nd, 0, sizeof(nd)); nd.kind = k; return nd; };
  // (a < 500 AND b >= 10) OR (a >= 3500 AND b < 5) OR (b == 63 AND a IN-range via two leaves)
  ips_node prog[] = {leaf(0, IPS_OP_LT, 500), leaf(1, IPS_OP_GE, 10), inner(IPS_NODE_AND),
                     leaf(0, IPS_OP_GE, 3500), leaf(1, IPS_OP_LT, 5), inner(IPS_NODE_AND),
                     inner(IPS_NODE_OR),
                     leaf(1, IPS_OP_EQ, 63), leaf(0, IPS_OP_GE, 1000), leaf(0, IPS_OP_LE, 2000),
                     inner(IPS_NODE_AND), inner(IPS_NODE_AND), inner(IPS_NODE_OR)};
  const int64_t words = (n + 63) / 64;
  ips::DeviceBuffer bm((size_t)words * 8);
  const int n_prog = (int)(sizeof(prog) / sizeof(prog[0]));
  const size_t ws_bytes = ips_program_workspace_bytes(prog, n_prog, cols, 2, n);
  CHECK(ws_bytes >= (size_t)words * 8);  // (A and B) or (C and D) or ...: two bitmaps alive
  CHECK(ips_eval_program(prog, n_prog, cols, 2, n, bm.as<uint64_t>(), nullptr, nullptr) == IPS_ERR_INVALID_ARG);
  ips::DeviceBuffer ws(ws_bytes);
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(ips_eval_program(prog, n_prog, cols, 2, n, bm.as<uint64_t>(), ws.get(), nullptr) == IPS_OK);
    std::vector<uint64_t> h((size_t)words);
    CHECK(bm.download(h.data(), (size_t)words * 8));
    int bad = 0, first_bad = -1;
    for (int i = 0; i < n; ++i) {
      const bool e = (a[i] < 500 && b[i] >= 10) || (a[i] >= 3500 && b[i] < 5) ||
                     (b[i] == 63 && a[i] >= 1000 && a[i] <= 2000);
      if ((((h[(size_t)i >> 6]) >> (i & 63)) & 1ull) != (uint64_t)e) {
        if (first_bad < 0) first_bad = i;
        ++bad;
      }
    }
    if (bad) fprintf(stderr, "program with temporary bitmap: rep %d: %d wrong bits, first at row %d (a=%u b=%u)\n",
                     rep, bad, first_bad, a[(size_t)first_bad], b[(size_t)first_bad]);
    CHECK(bad == 0);
  }
}

int main() {
  int count = 0;
  if (ips_device_count(&count) != IPS_OK || count == 0) {
    fprintf(stderr, "facade_test: no GPU: %s\n", ips_last_error());
    return 2;
  }
  TestFleSpecificSequences();
  TestFleValues();
  TestFlePredicates();
  TestNumbersAll<int8_t>(); TestNumbersAll<int16_t>(); TestNumbersAll<int32_t>();
  TestNumbersAll<int64_t>(); TestNumbersAll<float>(); TestNumbersAll<double>();
  TestDictPredicates<int32_t>(); TestDictPredicates<int64_t>(); TestDictPredicates<double>();
  TestDictPredicates<int16_t>(); TestDictPredicates<float>();
  TestScanner();
  TestScannerMultiPage();
  TestProgramWithTemporaryBitmap();
  TestCallPatternLaunchCounts();
  TestTruncatedPages();
  TestColumnChunkStream();
  TestAssembleRowsFused();
  TestLongInList();
  TestAssembleRowsChunks();
  CHECK(ips::sticky_status() == IPS_OK);
  printf("facade_test: %d checks, %d failed\n", g_checks, g_fail);
  return g_fail ? 1 : 0;
}
