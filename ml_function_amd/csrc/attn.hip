// A4  AutoInt interacting layer: entry points, forward kernels, reduce kernel (the code is csrc/attn_impl.h; the backward's
// instantiations compile in attn_bwd_f16.hip / attn_bwd_f32.hip).
#define FIL_ATTN_PART 0
#include "attn_impl.h"
