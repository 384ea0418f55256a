#define FVSRN_CD 5
#define FVSRN_PART 0
#include "kernels_inst.inc"
