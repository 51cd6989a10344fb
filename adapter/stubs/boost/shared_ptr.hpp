// STUB for the compile check of adapter/ (the image has no Boost): boost::shared_ptr with the part of its surface the adapter and
// the reference's drivers use.  In a LARVIO tree the real header is used instead.
#pragma once
#include <memory>
namespace boost { template <class T> using shared_ptr = std::shared_ptr<T>; }
