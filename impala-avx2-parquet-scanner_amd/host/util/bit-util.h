// util/bit-util.h (facade) -- the two BitUtil functions the scan path uses (bit-util.h:30-32,
// 128-140): Ceil and Log2 (= ceil(log2 x)), from which code bit widths are derived.
#pragma once
#include <stdint.h>

#include "../../../include/ips.h"

namespace impala {
class BitUtil {
 public:
  static inline int64_t Ceil(int64_t value, int64_t divisor) {
    return value / divisor + (value % divisor != 0);
  }
  static inline int Log2(uint64_t x) { return x <= 1 ? 0 : ips_dict_bit_width((int64_t)x); }
};
}  // namespace impala
