// <boost/shared_ptr.hpp> — STAND-IN (oracle/ref_shim/README.md): boost::shared_ptr is std::shared_ptr here.
#ifndef LINS_REF_SHIM_BOOST_SHARED_PTR_
#define LINS_REF_SHIM_BOOST_SHARED_PTR_
#include <memory>
namespace boost {
using std::shared_ptr;
}
#endif
