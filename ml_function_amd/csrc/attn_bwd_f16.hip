// the AutoInt backward kernel's f16-MFMA instantiations (csrc/attn_impl.h)
#define FIL_ATTN_PART 1
#include "attn_impl.h"
