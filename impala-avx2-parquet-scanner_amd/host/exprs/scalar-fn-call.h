// exprs/scalar-fn-call.h (MI355X facade) -- the plan-time lowering of conjuncts to the
// SimplePredicate tree: Expr::CreateSimplePredicates (expr.h:234-236, default NULL),
// ExprContext::CreateSimplePredicates (expr-context.h:129, expr-context.cc:340-342),
// ScalarFnCall::CreateSimplePredicates / CreateOperate (scalar-fn-call.cc:733-965) and
// And/OrPredicate::CreateSimplePredicates (compound-predicates.cc:36-67).
//
// Only the shapes the reference accepts are lowered: slot OP literal, cast(slot) OP literal,
// slot IN (literals), with fn names eq/gt/lt/ge/le/in_set_lookup; slot-vs-slot, BOOLEAN literals
// and unknown functions return NULL, which makes the scanner drop the vectorised path
// (hdfs-parquet-scanner.cc:1829-1832).  The template type comes from the LITERAL's type
// (scalar-fn-call.cc:743, quirk Q9).  Impala's Expr machinery itself (codegen, UDFs, row-wise
// GetBooleanVal) is out of scope; these classes carry just enough structure for the lowering.
#pragma once
#include <string>
#include <vector>

#include "../exec/hdfs-parquet-scanner.h"

namespace impala {

enum PrimitiveType { TYPE_BOOLEAN, TYPE_TINYINT, TYPE_SMALLINT, TYPE_INT, TYPE_BIGINT, TYPE_FLOAT,
                     TYPE_DOUBLE };

class Expr {
 public:
  explicit Expr(PrimitiveType t) : type_(t) {}
  virtual ~Expr() { for (Expr* c : children_) delete c; }
  // default: not expressible as a SimplePredicate, expr.h:234-236
  virtual SimplePredicate* CreateSimplePredicates(HdfsParquetScanner* /*scanner*/) { return NULL; }
  virtual bool is_slotref() const { return false; }
  PrimitiveType type() const { return type_; }
  void AddChild(Expr* e) { children_.push_back(e); }
  std::vector<Expr*> children_;

 protected:
  PrimitiveType type_;
};

class SlotRef : public Expr {
 public:
  SlotRef(PrimitiveType t, int slot_idx) : Expr(t), slot_idx_(slot_idx) {}
  virtual bool is_slotref() const { return true; }
  int slot_idx() const { return slot_idx_; }
 private:
  int slot_idx_;
};

class Cast : public Expr {  // cast(slot): children_[0] is the SlotRef (scalar-fn-call.cc:736-738)
 public:
  Cast(PrimitiveType t, Expr* child) : Expr(t) { AddChild(child); }
};

class Literal : public Expr {
 public:
  Literal(PrimitiveType t, double v) : Expr(t), i_((int64_t)v), d_(v) {}
  Literal(PrimitiveType t, int64_t v) : Expr(t), i_(v), d_((double)v) {}
  int64_t int_val() const { return i_; }
  double double_val() const { return d_; }
 private:
  int64_t i_;
  double d_;
};

class ScalarFnCall : public Expr {
 public:
  ScalarFnCall(const std::string& function_name, Expr* lhs, std::vector<Expr*> rhs)
      : Expr(TYPE_BOOLEAN), function_name_(function_name) {
    AddChild(lhs);
    for (Expr* e : rhs) AddChild(e);
  }

  virtual SimplePredicate* CreateSimplePredicates(HdfsParquetScanner* scanner) {
    SlotRef* slotref = NULL;
    if (children_[0]->children_.size() == 1 && children_[0]->children_[0]->is_slotref())
      slotref = static_cast<SlotRef*>(children_[0]->children_[0]);
    if (children_[0]->is_slotref()) slotref = static_cast<SlotRef*>(children_[0]);
    if (!slotref) return NULL;
    if (children_.size() < 2 || children_[1]->is_slotref()) return NULL;
    switch (children_[1]->type()) {
      case TYPE_BOOLEAN: return NULL;
      case TYPE_TINYINT: return CreateOperate<int8_t>(scanner, slotref);
      case TYPE_SMALLINT: return CreateOperate<int16_t>(scanner, slotref);
      case TYPE_INT: return CreateOperate<int32_t>(scanner, slotref);
      case TYPE_BIGINT: return CreateOperate<int64_t>(scanner, slotref);
      case TYPE_FLOAT: return CreateOperate<float>(scanner, slotref);
      case TYPE_DOUBLE: return CreateOperate<double>(scanner, slotref);
    }
    return NULL;
  }

 private:
  template <typename T>
  SimplePredicate* CreateOperate(HdfsParquetScanner* scanner, SlotRef* slotref) {
    std::vector<T> vals;
    for (size_t i = 1; i < children_.size(); ++i) {
      Literal* l = static_cast<Literal*>(children_[i]);
      vals.push_back(std::is_floating_point<T>::value ? (T)l->double_val() : (T)l->int_val());
    }
    const int slot_idx = slotref->slot_idx();
    SimplePredicate* operate = NULL;
    const bool binary = children_.size() == 2;
    if (function_name_ == "eq") { if (binary) operate = new EqOperate<T>(slot_idx, vals[0]); }
    else if (function_name_ == "gt") { if (binary) operate = new GtOperate<T>(slot_idx, vals[0]); }
    else if (function_name_ == "lt") { if (binary) operate = new LtOperate<T>(slot_idx, vals[0]); }
    else if (function_name_ == "ge") { if (binary) operate = new GeOperate<T>(slot_idx, vals[0]); }
    else if (function_name_ == "le") { if (binary) operate = new LeOperate<T>(slot_idx, vals[0]); }
    else if (function_name_ == "in_set_lookup") operate = new InOperate<T>(slot_idx, vals);
    return operate ? scanner->Own(operate) : NULL;
  }
  std::string function_name_;
};

template <int KIND>
class CompoundPredicate : public Expr {  // AndPredicate / OrPredicate
 public:
  CompoundPredicate(Expr* a, Expr* b) : Expr(TYPE_BOOLEAN) { AddChild(a); AddChild(b); }
  virtual SimplePredicate* CreateSimplePredicates(HdfsParquetScanner* scanner) {
    SimplePredicate* child0 = children_[0]->CreateSimplePredicates(scanner);
    if (!child0) return NULL;
    SimplePredicate* child1 = children_[1]->CreateSimplePredicates(scanner);
    if (!child1) return NULL;
    return scanner->Own(new BinaryOperate<KIND>(child0, child1));
  }
};
typedef CompoundPredicate<IPS_NODE_AND> AndPredicate;
typedef CompoundPredicate<IPS_NODE_OR> OrPredicate;

class ExprContext {
 public:
  explicit ExprContext(Expr* root) : root_(root) {}
  ~ExprContext() { delete root_; }
  SimplePredicate* CreateSimplePredicates(HdfsParquetScanner* scanner) {
    return root_->CreateSimplePredicates(scanner);  // expr-context.cc:340-342
  }
 private:
  Expr* root_;
};

// HdfsParquetScanner::CreateSimplePredicates, hdfs-parquet-scanner.cc:1825-1835: any conjunct
// that cannot be lowered clears the list (=> the caller must use the row-at-a-time path).
inline bool CreateSimplePredicates(HdfsParquetScanner* scanner, std::vector<ExprContext*>& conjunct_ctxs,
                                   std::vector<SimplePredicate*>* roots) {
  roots->clear();
  for (ExprContext* ctx : conjunct_ctxs) {
    SimplePredicate* root = ctx->CreateSimplePredicates(scanner);
    if (!root) { roots->clear(); return false; }
    roots->push_back(root);
  }
  return true;
}

}  // namespace impala
