/* MOCK (see mlvalues.h) */
#ifndef MOCK_CAML_CALLBACK_H
#define MOCK_CAML_CALLBACK_H
#include "mlvalues.h"
value caml_callback(value closure, value arg);
value caml_callback_exn(value closure, value arg);
value caml_callback3_exn(value closure, value a, value b, value c);
#endif
