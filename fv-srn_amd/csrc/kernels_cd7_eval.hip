#define FVSRN_CD 7
#define FVSRN_PART 0
#include "kernels_inst.inc"
