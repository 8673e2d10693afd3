/* MOCK of the OCaml runtime's C interface -- declarations only, for `gcc -fsyntax-only` of bindings/ocaml/ptx_stubs.c in an
 * image that has no OCaml (tests/test_ocaml_binding.py).  Just enough of the documented API ("Interfacing C with OCaml",
 * OCaml manual ch. 22) for the compiler to check names, arities and types; nothing here is ever linked or run. */
#ifndef MOCK_CAML_MLVALUES_H
#define MOCK_CAML_MLVALUES_H
#include <stddef.h>
#include <stdint.h>
typedef intptr_t intnat;
typedef uintptr_t uintnat;
typedef intnat value;
typedef uintnat mlsize_t;
#define CAMLprim
#define Val_long(x) ((value)(((uintnat)(intnat)(x) << 1) + 1))
#define Long_val(x) ((intnat)(x) >> 1)
#define Val_int(x) Val_long(x)
#define Int_val(x) ((int)Long_val(x))
#define Val_unit Val_long(0)
#define Field(x, i) (((value*)(x))[i])
#define Hd_val(v) (((uintnat*)(v))[-1])
#define Wosize_val(v) ((mlsize_t)(Hd_val(v) >> 10))
#define Double_wosize ((mlsize_t)(sizeof(double) / sizeof(value)))
#define Is_exception_result(v) (((v) & 3) == 2)
#define Extract_exception(v) ((v) & ~(value)3)
#endif
