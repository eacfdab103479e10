// The backward layer kernel with z3 = SiLU(z2) Wc1^T + bc1 RECOMPUTED per tile (IMMUNOSTRUCT_SAVE_Z3=0): the same source as
// egnn_layer_bwd.hip, compiled with IS_BWD_Z3R = 1 into kernels / a launcher of their own (is::launch_layer_bwd_z3r, called by
// is_egnn_layer_bwd when the forward did not save z3).  A translation unit of its own so that the default build never sees this
// form: as a template parameter it changed the default instantiations' register allocation (HISTORY.md).
#define IS_BWD_Z3R 1
#include "egnn_layer_bwd.hip"
