#define FVSRN_CD 1
#define FVSRN_PART 6
#include "kernels_inst.inc"
