#define FVSRN_CD 8
#define FVSRN_PART 2
#include "kernels_inst.inc"
