// stand-in: see opencv.hpp (test infrastructure only)
#pragma once
#include "opencv.hpp"
