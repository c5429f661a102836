// pipeline.hip -- bin pipeline (placeholder until the fp64 kernels land).
#include "pipeline.hpp"

namespace bf {
BinPipeline *BinPipeline::create(const bf_config &, int) { return nullptr; }
}  // namespace bf
