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

#ifdef FD_PW_PROBE
// measurement aid (tools/pw_bwd_phases.py): select the unit whose paired pointwise backward launch is stamped, read the stamps back
extern "C" int fd_pw_probe_select(int M, int N, int K)
{
    const int sel[3] = {M, N, K};
    if (hipMemcpyToSymbol(HIP_SYMBOL(fd_pw_probe_sel), sel, sizeof sel) != hipSuccess) return -1;
    static long long zero[8 * FD_PW_PROBE_SLOTS];
    return hipMemcpyToSymbol(HIP_SYMBOL(fd_pw_probe), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
extern "C" int fd_pw_probe_read(long long *host, int slots)
{
    if (slots > FD_PW_PROBE_SLOTS) slots = FD_PW_PROBE_SLOTS;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(fd_pw_probe), (size_t)slots * 8 * sizeof(long long)) == hipSuccess ? slots : -1;
}
#endif
