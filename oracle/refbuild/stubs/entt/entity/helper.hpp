// see registry.hpp (EnTT stand-in, test infrastructure)
#pragma once
#include "registry.hpp"
