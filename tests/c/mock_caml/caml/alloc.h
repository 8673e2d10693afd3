/* MOCK (see mlvalues.h) */
#ifndef MOCK_CAML_ALLOC_H
#define MOCK_CAML_ALLOC_H
#include "mlvalues.h"
value caml_alloc_tuple(mlsize_t);
value caml_copy_double(double);
value caml_copy_string(const char*);
#endif
