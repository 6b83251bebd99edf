// fd_train_fwd.hip -- translation unit 2 of libfastdepth_hip.so: train plan creation, train-mode forward, the train plan's test hooks, the
// evaluation-input gather (fd_val_transform).  Compiled in parallel with fd_api.hip (inference) and fd_train_bwd.hip (backward, loss, SGD, exchange).
#include "fd_kernels_f32.h"          // (FD_F32_STAGES: fd_plan_select.h)
#include "fd_kernels_train.h"
#include "fd_kernels_gemm16_f32.h"
#include "fd_kernels_train_h16.h"
#include "fd_kernels_dw5p_bwd.h"
#include "fd_kernels_io.h"
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
#include "fd_plan_select.h"   // choose_pw16: the fp32 plan's forward GEMMs take the inference plan's one-round rule
#include "fd_train_impl.h"
