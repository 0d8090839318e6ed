// exprs/simple-predicates.h (MI355X facade) -- the vectorised predicate tree of the reference
// (simple-predicates.h:30-205): SimplePredicate::GetBitset(scanner, num_rows, bitset) with
// AndOperate / OrOperate and the leaves {Eq,Lt,Le,Gt,Ge,In}Operate<T>(slot_idx, literal[s]).
//
// Two ways to evaluate a tree:
//   GetBitset()  node by node, exactly the reference's call pattern (each leaf calls
//                scanner->Eq/Lt/...; And/Or combine temporaries, simple-predicates.h:145-163)
//   Lower()      emit the tree as a postfix ips_node program; the scanner facade hands the whole
//                program to ips_eval_program, which reads every column once and writes one bitmap
#pragma once
#include <vector>

#include "../ips/runtime.h"

namespace impala {

using ips::SkipBitset;
class HdfsParquetScanner;

class SimplePredicate {
 public:
  virtual void GetBitset(HdfsParquetScanner* scanner, int64_t num_rows, SkipBitset& skip_bitset) = 0;
  // facade extra: append this subtree in postfix order; false if it cannot be expressed
  virtual bool Lower(HdfsParquetScanner* scanner, std::vector<ips_node>* program) = 0;
  virtual ~SimplePredicate() {}
};

// AND / OR of two subtrees (AndOperate / OrOperate in the reference)
template <int KIND>
class BinaryOperate : public SimplePredicate {
 public:
  BinaryOperate(SimplePredicate* child0, SimplePredicate* child1) : child0_(child0), child1_(child1) {}
  virtual void GetBitset(HdfsParquetScanner* scanner, int64_t num_rows, SkipBitset& skip_bitset) {
    child0_->GetBitset(scanner, num_rows, skip_bitset);
    SkipBitset tmp_bitset;
    child1_->GetBitset(scanner, num_rows, tmp_bitset);
    if (KIND == IPS_NODE_AND) skip_bitset &= tmp_bitset; else skip_bitset |= tmp_bitset;
  }
  virtual bool Lower(HdfsParquetScanner* scanner, std::vector<ips_node>* program) {
    if (!child0_->Lower(scanner, program) || !child1_->Lower(scanner, program)) return false;
    ips_node n;
    memset(&n, 0, sizeof(n));
    n.kind = KIND;
    program->push_back(n);
    return true;
  }

 private:
  SimplePredicate* child0_;
  SimplePredicate* child1_;
};
typedef BinaryOperate<IPS_NODE_AND> AndOperate;
typedef BinaryOperate<IPS_NODE_OR> OrOperate;

// slot OP literal(s); the member templates of HdfsParquetScanner do the work
template <typename T, int OP>
class LeafOperate : public SimplePredicate {
 public:
  LeafOperate(int idx, T val) : idx_(idx), vals_(1, val) {}
  LeafOperate(int idx, std::vector<T> vals) : idx_(idx), vals_(vals.begin(), vals.end()) {}
  virtual void GetBitset(HdfsParquetScanner* scanner, int64_t num_rows, SkipBitset& skip_bitset);
  virtual bool Lower(HdfsParquetScanner* scanner, std::vector<ips_node>* program);

 private:
  int idx_;
  std::vector<T> vals_;
};
template <typename T> using EqOperate = LeafOperate<T, IPS_OP_EQ>;
template <typename T> using LtOperate = LeafOperate<T, IPS_OP_LT>;
template <typename T> using LeOperate = LeafOperate<T, IPS_OP_LE>;
template <typename T> using GtOperate = LeafOperate<T, IPS_OP_GT>;
template <typename T> using GeOperate = LeafOperate<T, IPS_OP_GE>;
template <typename T> using InOperate = LeafOperate<T, IPS_OP_IN>;

}  // namespace impala
