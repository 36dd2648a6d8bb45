// lins_ctx_priv.h — what other translation units of liblins_ieskf.so may use of a lins_ctx (the struct itself
// is private to lins_capi.hip): its stream / device, the error-string slot, and one attachment slot for the
// state of the scan-to-map row (freed by lins_destroy through the registered function).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/lins_ieskf.h"

namespace lins {
hipStream_t ctx_stream(lins_ctx* ctx);
int ctx_device(lins_ctx* ctx);
int ctx_fail_hip(lins_ctx* ctx, hipError_t e, const char* what);  // records the message, returns LINS_E_HIP
void** ctx_map_slot(lins_ctx* ctx, void (*free_fn)(void*));       // attachment slot (free_fn is remembered)
// the context's event pair for kernel timing
void ctx_events(lins_ctx* ctx, hipEvent_t* a, hipEvent_t* b);
const lins_params* ctx_params(const lins_ctx* ctx);  // the parameters the context was created with
}  // namespace lins
