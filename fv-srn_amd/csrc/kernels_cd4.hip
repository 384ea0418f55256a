#define FVSRN_CD 4
#include "kernels_inst.inc"
