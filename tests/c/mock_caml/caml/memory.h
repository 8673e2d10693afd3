/* MOCK (see mlvalues.h): GC-root bookkeeping macros reduced to declarations that type-check their arguments. */
#ifndef MOCK_CAML_MEMORY_H
#define MOCK_CAML_MEMORY_H
#include "mlvalues.h"
void mock_caml_root(value*);
#define CAMLparam0() int caml__frame_ = 0; (void)caml__frame_
#define CAMLparam1(a) CAMLparam0(); mock_caml_root(&(a))
#define CAMLparam2(a, b) CAMLparam1(a); mock_caml_root(&(b))
#define CAMLparam3(a, b, c) CAMLparam2(a, b); mock_caml_root(&(c))
#define CAMLparam4(a, b, c, d) CAMLparam3(a, b, c); mock_caml_root(&(d))
#define CAMLparam5(a, b, c, d, e) CAMLparam4(a, b, c, d); mock_caml_root(&(e))
#define CAMLxparam1(a) mock_caml_root(&(a))
#define CAMLxparam2(a, b) CAMLxparam1(a); mock_caml_root(&(b))
#define CAMLxparam3(a, b, c) CAMLxparam2(a, b); mock_caml_root(&(c))
#define CAMLlocal1(a) value a = Val_unit; mock_caml_root(&(a))
#define CAMLlocal2(a, b) CAMLlocal1(a); CAMLlocal1(b)
#define CAMLlocal3(a, b, c) CAMLlocal2(a, b); CAMLlocal1(c)
#define CAMLreturn(x) do { (void)caml__frame_; return (x); } while (0)
#define CAMLreturn0 do { (void)caml__frame_; return; } while (0)
#define CAMLdrop ((void)caml__frame_)
void caml_modify(value*, value);
/* generational global roots (memory.h of the runtime): the cell must not move while it is registered */
void caml_register_generational_global_root(value*);
void caml_remove_generational_global_root(value*);
void caml_modify_generational_global_root(value*, value);
#define Store_field(block, i, v) caml_modify(&Field(block, i), v)
#endif
