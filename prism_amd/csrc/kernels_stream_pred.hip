// The streaming GROUP BY kernels that evaluate WHERE themselves: the same source as
// kernels_stream.hip, instantiated with PRED = true, as a translation unit of its own so
// that the two halves compile side by side.
#define PRISM_STREAM_PRED_TU 1
#include "kernels_stream.hip"
