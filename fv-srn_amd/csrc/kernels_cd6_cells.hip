#define FVSRN_CD 6
#define FVSRN_PART 5
#include "kernels_inst.inc"
