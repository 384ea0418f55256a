#define FVSRN_CD 7
#define FVSRN_PART 2
#include "kernels_inst.inc"
