// what conv3x3.hip expects from the rest of the library, for the stand-alone diagnostic build
#include "hd_common.h"
extern "C" int hd_bn_act_replicas(void) { return 16; }
int hd_conv3x3_wgrad_supported_v1(int, int, int, int, int, int) { return 0; }
size_t hd_conv3x3_wgrad_workspace_bytes_v1(int, int, int, int, int, int) { return 0; }
hd_status hd_conv3x3_wgrad_partial_v1(const void*, int64_t, const void*, int64_t, int, int, int, int, int, int, void*, void*, size_t,
                                      const hd_wgrad_job*, hd_wgrad_job*, hd_stream) { return HD_ERR_SHAPE; }
extern "C" hd_status hd_wgrad_reduce(const hd_wgrad_job*, hd_stream) { return HD_OK; }
