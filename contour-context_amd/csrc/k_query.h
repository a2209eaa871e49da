// query kernels (KNN / constellation checks / GMM) -- filled in below
#pragma once
#include "cc_dev.h"
