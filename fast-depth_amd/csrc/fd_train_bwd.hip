// fd_train_bwd.hip -- translation unit 3 of libfastdepth_hip.so: backward pass of the train plan, L1 loss, depth metrics, fused SGD, gradient casts and the
// library-issued RCCL exchange.  Compiled in parallel with fd_api.hip (inference) and fd_train_fwd.hip (plan creation + forward).
#include "fd_kernels_train.h"
#include "fd_kernels_bwd.h"
#include "fd_kernels_train_h16.h"
#include "fd_kernels_dw5p_bwd.h"
#include "../../include/fastdepth_hip.h"
#include "fd_tuning.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#ifndef FD_EMU
#include <hip/hip_ext.h>
#endif

#include "fd_host_common.h"
#include "fd_train_bwd_impl.h"
