(* ptx.ml -- OCaml side of the binding to libptx_hip.so (include/ptx.h): the MI355X path integrator behind
   Integrator.create / Integrator.render.

   The reference hands the integrator three closures (camera, intersect, background: render_command.mli:18-22);
   a GPU cannot call closures, so this module takes the same scene DECLARATIVELY: the spheres main.ml has just
   built (already in camera space), what their materials and textures are, the four numbers Camera.ray reads,
   and the background gradient.  [render] then fills the Bimage exactly as Integrator.render would. *)
open! Base

module Texture = struct
  type rgb = float * float * float

  type t =
    | Solid of rgb (* Texture.solid *)
    | Checker of
        { width : int
        ; height : int
        ; even : rgb
        ; odd : rgb
        } (* Texture.checker ~width ~height (solid even) (solid odd) *)
end

module Material = struct
  type t =
    | Lambertian of Texture.t
    | Metal of Texture.t
    | Dielectric of float (* index; Material.glass = Dielectric 1.5 *)
end

type sphere =
  { x : float
  ; y : float
  ; z : float (* centre in CAMERA space: Sphere.transform s ~f:(Camera.transform camera) *)
  ; radius : float
  ; material : Material.t
  }

type camera =
  { lower_left_x : float
  ; lower_left_y : float
  ; view_x : float
  ; view_y : float
  }

type background =
  | Black
  | Sky of
      { horizon : Texture.rgb
      ; zenith : Texture.rgb
      } (* lerp t horizon zenith, t = .5 * (normalize(dir).y + 1): shirley_spheres/bin/main.ml:104-110 *)

type leaf =
  | Simd_leaf (* <= leaf_size () spheres per packet, the Rust x86 arithmetic *)
  | Array_leaf of int (* --no-simd: Sphere.intersect, length_cutoff *)

(* what crosses the FFI: flat unboxed arrays, like Simd_leaf.coords (main.ml:137-142) *)
type flat =
  { xs : floatarray
  ; ys : floatarray
  ; zs : floatarray
  ; rs : floatarray
  ; sphere_material : (int32, Bigarray.int32_elt, Bigarray.c_layout) Bigarray.Array1.t
  ; materials : floatarray (* 6 per material: kind, texture, index, emit r g b *)
  ; textures : floatarray (* 9 per texture: kind, width, height, even r g b, odd r g b *)
  ; camera : floatarray (* lower_left_x, lower_left_y, view_x, view_y *)
  ; background : floatarray (* kind, horizon r g b, zenith r g b *)
  ; leaf_kind : int
  ; length_cutoff : int
  }

type scene (* custom block around the ptx_scene* handle; finalised by the GC *)

external leaf_size : unit -> int = "ptx_ml_leaf_size"
external device_count : unit -> int = "ptx_ml_device_count"
external scene_create_flat : flat -> int -> scene = "ptx_ml_scene_create_stub"
external scene_destroy : scene -> unit = "ptx_ml_scene_destroy_stub"

external render_flat
  :  scene
  -> int (* width *)
  -> int (* height *)
  -> int (* samples_per_pixel *)
  -> int (* max_bounces *)
  -> int (* gpus *)
  -> (float, Bigarray.float64_elt, Bigarray.c_layout) Bigarray.Array1.t
  -> (int -> unit)
  -> unit
  = "ptx_ml_render_bytecode" "ptx_ml_render"

module FA = Stdlib.Float.Array

let flatten ~leaf camera background (spheres : sphere array) =
  let n = Array.length spheres in
  let xs = FA.init n (fun i -> spheres.(i).x)
  and ys = FA.init n (fun i -> spheres.(i).y)
  and zs = FA.init n (fun i -> spheres.(i).z)
  and rs = FA.init n (fun i -> spheres.(i).radius) in
  (* one material and at most one texture per sphere, in sphere order: no de-duplication, the tables are tiny *)
  let textures = Queue.create ()
  and materials = Queue.create () in
  let add_texture (t : Texture.t) =
    let row =
      match t with
      | Solid (r, g, b) -> [ 0.; 0.; 0.; r; g; b; 0.; 0.; 0. ]
      | Checker { width; height; even = er, eg, eb; odd = or_, og, ob } ->
        [ 1.; Float.of_int width; Float.of_int height; er; eg; eb; or_; og; ob ]
    in
    let idx = Queue.length textures / 9 in
    List.iter row ~f:(Queue.enqueue textures);
    idx
  in
  let sphere_material = Bigarray.Array1.create Bigarray.int32 Bigarray.c_layout n in
  Array.iteri spheres ~f:(fun i s ->
    let row =
      match s.material with
      | Lambertian t -> [ 0.; Float.of_int (add_texture t); 0.; 0.; 0.; 0. ]
      | Metal t -> [ 1.; Float.of_int (add_texture t); 0.; 0.; 0.; 0. ]
      | Dielectric index -> [ 2.; 0.; index; 0.; 0.; 0. ]
    in
    List.iter row ~f:(Queue.enqueue materials);
    sphere_material.{i} <- Int32.of_int_exn i);
  let of_queue q = FA.of_list (Queue.to_list q) in
  let background =
    match background with
    | Black -> FA.of_list [ 0.; 0.; 0.; 0.; 0.; 0.; 0. ]
    | Sky { horizon = hr, hg, hb; zenith = zr, zg, zb } -> FA.of_list [ 1.; hr; hg; hb; zr; zg; zb ]
  in
  let leaf_kind, length_cutoff =
    match leaf with
    | Simd_leaf -> 0, leaf_size ()
    | Array_leaf cutoff -> 1, cutoff
  in
  { xs
  ; ys
  ; zs
  ; rs
  ; sphere_material
  ; materials = of_queue materials
  ; textures = of_queue textures
  ; camera = FA.of_list [ camera.lower_left_x; camera.lower_left_y; camera.view_x; camera.view_y ]
  ; background
  ; leaf_kind
  ; length_cutoff
  }
;;

(* Shape_tree.create + uploading the scene: what main.ml does before Render_cmd.run *)
let scene_create ?(device = 0) ?(leaf = Simd_leaf) ~camera ~background spheres =
  scene_create_flat (flatten ~leaf camera background spheres) device
;;

(* Integrator.create ... |> Integrator.render ~update_progress, on [gpus] GPUs of this node *)
let render ?(gpus = 1) scene ~width ~height ~samples_per_pixel ~max_bounces ~image ~update_progress =
  render_flat scene width height samples_per_pixel max_bounces gpus image update_progress
;;
