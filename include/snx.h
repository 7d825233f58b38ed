/* temporary minimal header during bring-up; the documented C ABI is written once all ops exist */
#pragma once
#include <stdint.h>
#include <stddef.h>
#ifndef __HIP__
typedef struct ihipStream_t* hipStream_t;
#endif
