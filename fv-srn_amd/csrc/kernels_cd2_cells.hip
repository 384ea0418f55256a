#define FVSRN_CD 2
#define FVSRN_PART 5
#include "kernels_inst.inc"
