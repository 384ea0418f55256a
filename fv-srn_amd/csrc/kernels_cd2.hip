#define FVSRN_CD 2
#include "kernels_inst.inc"
