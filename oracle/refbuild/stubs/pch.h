// the reference units say #include "pch.h"; ref_pch.h (prefix header) already holds everything
#pragma once
#include "../ref_pch.h"
