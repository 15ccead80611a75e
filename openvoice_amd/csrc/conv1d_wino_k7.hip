// Winograd-domain conv instances, kernel size 7 (conv1d_wino.h): dilations 1 (one and two fragments per wave), 3, 5.
#include "conv1d_wino.h"

namespace ovkw { using namespace ovkw; }
using ovkw::wino_launch;
OVW_DEFINE_DISPATCH(7)
