// The two- and four-slice instances of pwdw.hip (PwDwArgs::sl = 2 | 4: the batched early MobileNet / SSD pairs) as a code object of
// their own -- see the note in pwdw.hip above pwdw_lds_bytes.
#define TAMD_PWDW_SLICES_TU 1
#include "pwdw.hip"
