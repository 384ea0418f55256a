#define FVSRN_CD 1
#define FVSRN_PART 1
#include "kernels_inst.inc"
