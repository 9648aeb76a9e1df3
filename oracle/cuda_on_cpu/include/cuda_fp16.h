#pragma once
#include "cuda_runtime.h"
struct __half { unsigned short x; }; struct __nv_bfloat16 { unsigned short x; };
