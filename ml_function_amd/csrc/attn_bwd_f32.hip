// the AutoInt backward kernel's exact-fp32 instantiations (csrc/attn_impl.h)
#define FIL_ATTN_PART 2
#include "attn_impl.h"
