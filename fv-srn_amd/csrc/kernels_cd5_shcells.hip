#define FVSRN_CD 5
#define FVSRN_PART 6
#include "kernels_inst.inc"
