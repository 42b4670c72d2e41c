// A4  AutoInt interacting layer -- placeholder entry points (implemented next).
#include "common.h"
using namespace fil;
extern "C" size_t fil_attn_fwd_workspace_bytes(int, int, int, int, int) { return 0; }
extern "C" size_t fil_attn_bwd_workspace_bytes(int, int, int, int, int) { return 0; }
extern "C" int fil_attn_fwd(const float*, const float*, const float*, const float*, const float*, const float*, float*, int, int,
                            int, int, int, float, float, void*, size_t, void*) {
  return fail(FIL_ERR_UNSUPPORTED, "fil_attn_fwd: not built yet");
}
extern "C" int fil_attn_bwd(const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                            float*, float*, float*, float*, float*, float*, int, int, int, int, int, float, float, void*, size_t,
                            void*) {
  return fail(FIL_ERR_UNSUPPORTED, "fil_attn_bwd: not built yet");
}
