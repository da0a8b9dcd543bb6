// oracle/ref_dropin_pathtrace.cpp — TEST INFRASTRUCTURE.  The reference's yocto_pathtrace translation unit with ONE
// function replaced: its pathtrace_samples keeps living under another name (nothing calls it), and the name
// yocto::pathtrace_samples is defined by the reference-side binding stub (ref_dropin_stub.h) over libvpt_hip.so.
// Linked with the reference's other objects and our headless driver (ref_driver.cpp, the run_offline sequence) this
// is oracle/_ref/ref_dropin: the reference application rendering through the HIP path.
#define pathtrace_samples pathtrace_samples_on_the_cpu
#include <yocto_pathtrace/yocto_pathtrace.cpp>   // from where it lies under /root/reference (oracle/Makefile: -I$(REF)/libs)
#undef pathtrace_samples

#include "ref_dropin_stub.h"
