/* MOCK (see mlvalues.h) */
#ifndef MOCK_CAML_FAIL_H
#define MOCK_CAML_FAIL_H
#include "mlvalues.h"
_Noreturn void caml_failwith(const char* msg);
_Noreturn void caml_invalid_argument(const char* msg);
_Noreturn void caml_raise(value exn);
#endif
