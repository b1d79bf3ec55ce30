// kernels_blind_rotate.hpp — every blind-rotation kernel family of the N = 1024 / 2048 engine, one header per family, in the order
// the translation units have always seen them (kernels_anyn.hpp and kernels_n512.hpp hold the other degrees).
#pragma once
#include "kernels_common.hpp"
#include "kernels_v3.hpp"
#include "kernels_multikey.hpp"
#include "kernels_w2.hpp"
#include "kernels_h2.hpp"
#include "kernels_k2.hpp"
#include "kernels_n2048.hpp"
#include "kernels_general.hpp"
#include "kernels_keyprep.hpp"
