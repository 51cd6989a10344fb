// stand-in: see lvref_cv.hpp (test infrastructure, lets /root/reference/src/ORBDescriptor.cpp compile in place)
#include "../lvref_cv.hpp"
