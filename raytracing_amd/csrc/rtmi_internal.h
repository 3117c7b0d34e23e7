// rtmi_internal.h -- hooks between the translation units of librtmi.so (not part of the C ABI, hidden visibility).
#pragma once
#include <cstdint>

#include "../../include/rtmi.h"

// sets rtmi_last_error() and returns `code`
int rtmi_internal_fail(int code, const char* msg);
// rtmi_isochrones with the result left on the device: *d_out = [ntimes][3][R] fp64 (caller hipFree's it), in the CALLER's
// ray order; *R and *stream (a hipStream_t) describe the batch.  The stream is synchronised when this returns.
int rtmi_internal_isochrones_device(rtmi_batch* b, int32_t ntimes, const double* times, double** d_out, long* R, void** stream);
// what 0: d_ray [3][R] (rtmi_read_d_ray's content), 1: final [9][R] (rtmi_read_final's), fp64, the caller's ray order, written to
// dst -- DEVICE memory on the batch's device -- by a kernel enqueued on `stream` (a hipStream_t) after the batch's own stream
// has been synchronised.  For shard.hip's device-to-device read-back.
int rtmi_internal_pack_device(rtmi_batch* b, int what, double* dst, void* stream);
