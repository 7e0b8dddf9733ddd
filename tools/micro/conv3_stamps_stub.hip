// what conv3x3.hip expects from the rest of the library, for the stand-alone diagnostic build
#include "hd_common.h"
extern "C" int hd_bn_act_replicas(void) { return 16; }
hd_status hd_conv3x3_fwd_v1(const void*, int64_t, const void*, int, int, int, int, int, void*, int64_t, float*, hd_stream) { return HD_ERR_SHAPE; }
