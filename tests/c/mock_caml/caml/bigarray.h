/* MOCK (see mlvalues.h) */
#ifndef MOCK_CAML_BIGARRAY_H
#define MOCK_CAML_BIGARRAY_H
#include "mlvalues.h"
struct caml_ba_array {
  void* data;
  intnat num_dims;
  intnat flags;
  void* proxy;
  intnat dim[1];
};
#define Data_custom_val(v) ((void*)&Field(v, 1))
#define Caml_ba_array_val(v) ((struct caml_ba_array*)Data_custom_val(v))
#define Caml_ba_data_val(v) (Caml_ba_array_val(v)->data)
#endif
