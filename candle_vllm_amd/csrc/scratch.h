// Device scratch registry (scratch.cpp): one buffer per (device, stream, key); growing retires instead of freeing.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

enum {
    MI355_SCR_PA_ARRIVE = 1,     // fused attention merge: arrival tickets (zero-initialised, self-cleaning)
    MI355_SCR_ARGMAX = 2,        // arg-max stage-1 slots (zero-initialised, self-cleaning)
    MI355_SCR_QMM_IMG0 = 3,      // wide mat-mul path: activation image, buffer 0 / 1
    MI355_SCR_QMM_IMG1 = 4,
    MI355_SCR_QMM_PART = 5,      // wide mat-mul path: split-K partial sums
    MI355_SCR_QMM_TICKET = 6,    // wide mat-mul path: split-K arrival tickets (zero-initialised, self-cleaning)
    MI355_SCR_QMP_WS = 7,        // prompt-step GEMM: quantised activation image
    MI355_SCR_DENSE_WS = 8,      // dense / GPTQ path staging
    MI355_SCR_COMM = 9,          // one-shot all-reduce staging
    MI355_SCR_GPTQ_WIDE = 10,    // 5..64-token launches of tiled 4-bit weights: split-K partial sums (gptq_wide.inc)
    MI355_SCR_PF_FP8 = 11,       // prompt attention over an e4m3 cache: the sequences' blocks as bf16 + their block table (prefill_attention.hip)
};

// *out = a device buffer of >= bytes owned by (current device, st, key).  zero_on_create: the buffer is cleared (on `st`)
// when it is first created or grown.  Returns hipErrorStreamCaptureUnsupported when growth would be needed mid-capture.
int mi355_scratch_get(void** out, int key, size_t bytes, hipStream_t st, bool zero_on_create);
extern "C" void mi355_scratch_release_all(void);

// void FFI entry points record their first failure here (mi355_last_error / mi355_clear_error in the header)
void mi355_note_error(int code);
