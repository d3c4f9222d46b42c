// w2c_option() for single-file debug builds of csrc/stem.hip (tools/stem_phases.py): the table lives in conv_igemm.hip in the library.
#include <cstdlib>
#include "../../multiagentperception_amd/csrc/w2c_common.h"
int w2c_option(int id) {
    static const char* const names[W2C_OPT_COUNT] = {"W2C_XCD2D", "W2C_NO_S2PATCH", "W2C_STEM_WGS", "W2C_STEM_FORM", "W2C_STEM_BAND",
                                                     "W2C_STEM_WAVES", "W2C_WGRAD_PATCH", "W2C_INWG_SPLITK"};
    static const int defaults[W2C_OPT_COUNT] = {1, 0, 0, 0, 8, 8, 1, 1};
    if (id < 0 || id >= W2C_OPT_COUNT) return 0;
    const char* e = getenv(names[id]);
    return e ? atoi(e) : defaults[id];
}
