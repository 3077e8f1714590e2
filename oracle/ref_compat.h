/* oracle/ref_compat.h -- TEST INFRASTRUCTURE.  Force-included (-include) when the
 * reference's own .cc files are compiled in place for oracle/_ref (see build_ref.sh).
 * The reference was written for gcc 4.8; system/System.cc:1133 returns a std::ifstream
 * where a bool is expected, which C++11 libstdc++ (explicit operator bool) rejects.
 * This shim makes the unqualified name `ifstream` a subclass that still converts
 * implicitly; it does not stand in for any reference header or library. */
#ifndef DFK_REF_COMPAT_H
#define DFK_REF_COMPAT_H
#ifdef __cplusplus
#include <fstream>
namespace std { namespace refcompat {
class ifstream : public std::ifstream
{ public: using std::ifstream::ifstream; operator bool() const { return !this->fail(); } };
} }
namespace refcompat = std::refcompat;
#define ifstream refcompat::ifstream
#endif
#endif
