#define FVSRN_CD 7
#define FVSRN_PART 5
#include "kernels_inst.inc"
