(* ptx.ml -- OCaml side of the binding to libptx_hip.so (include/ptx.h): the MI355X integrators behind
   Integrator.create / Integrator.render (shirley_spheres) and Progressive_photon_map.Make(Scene).go (cornell-box, ganesha).

   The reference hands its integrators closures (camera, intersect, background: render_command.mli:18-22;
   progressive_photon_map.mli:35-41); a GPU cannot call closures, so this module takes the same scene DECLARATIVELY:
   the spheres / triangle mesh / pre-tested floor triangles main.ml has just built (already in camera space), what their
   materials and textures are, the four numbers Camera.ray reads, the background gradient, and -- for the photon mapper --
   the lights.  [render] then fills the Bimage exactly as Integrator.render would; [ppm_render] accumulates img_sum
   exactly as Progressive_photon_map's [go] does and calls back after every iteration so the caller can save the image. *)
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
    | Emitting of t * Texture.rgb
        (* the Hit.emit slot (hit.ml:5): the reference's Material.emit is constant black (material.ml:59); non-black values
           are the documented extension that lights cornell-box under the path integrator *)
end

type sphere =
  { x : float
  ; y : float
  ; z : float (* centre in CAMERA space: Sphere.transform s ~f:(Camera.transform camera) *)
  ; radius : float
  ; material : Material.t
  }

type uv = float * float (* Texture.Coord.t *)

(* Triangle.Make(Face).t over shared vertices (ganesha's Mesh.t + Face.t, ganesha/bin/main.ml:37-119) or with private
   vertices (cornell-box's Face.t, cornell-box/bin/main.ml:5-28: three fresh vertices per face) *)
type mesh =
  { vx : floatarray
  ; vy : floatarray
  ; vz : floatarray (* vertices in CAMERA space *)
  ; faces : (int32, Bigarray.int32_elt, Bigarray.c_layout) Bigarray.Array1.t (* 3 vertex indices per triangle: a, b, c *)
  ; face_uv : uv * uv * uv (* Face.tex_coords of face i = face_uvs.(i) if given, else this *)
  ; face_uvs : (uv * uv * uv) array (* empty = every face uses [face_uv] *)
  ; face_material : Material.t (* Face.material of face i = face_materials.(i) if given, else this *)
  ; face_materials : Material.t array (* empty = every face uses [face_material] *)
  }

let empty_mesh =
  { vx = Stdlib.Float.Array.create 0
  ; vy = Stdlib.Float.Array.create 0
  ; vz = Stdlib.Float.Array.create 0
  ; faces = Bigarray.Array1.create Bigarray.int32 Bigarray.c_layout 0
  ; face_uv = (0., 0.), (0., 0.), (0., 0.)
  ; face_uvs = [||]
  ; face_material = Material.Dielectric 1.5
  ; face_materials = [||]
  }
;;

type p3 = float * float * float

(* a triangle tested BEFORE the tree, its hit clipping t_max for the tree (ganesha's Floor, ganesha/bin/main.ml:205-298) *)
type floor_triangle =
  { vertices : p3 * p3 * p3
  ; uvs : uv * uv * uv
  ; surface : Material.t
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
  | Array_leaf of int (* Shape_tree.Array_leaf, length_cutoff: 4 (--no-simd), 2 (cornell-box), 8 (ganesha) *)

(* Progressive_photon_map.Light.create_point / create_spot (progressive_photon_map.ml:59-137), positions in camera space *)
type light =
  | Point of
      { position : p3
      ; color : Texture.rgb
      ; power : float
      }
  | Spot of
      { position : p3
      ; direction : p3 (* not normalised by the caller, like create_spot's ~direction *)
      ; color : Texture.rgb
      ; power : float
      }

(* what crosses the FFI: flat unboxed arrays, like Simd_leaf.coords (main.ml:137-142).  The stub indexes this record
   with Field (flat, i): keep the field order in step with ptx_stubs.c *)
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
  ; vertex_x : floatarray
  ; vertex_y : floatarray
  ; vertex_z : floatarray
  ; tri_indices : (int32, Bigarray.int32_elt, Bigarray.c_layout) Bigarray.Array1.t (* 3 per triangle *)
  ; tri_uv : floatarray (* 6 per triangle: ua va ub vb uc vc *)
  ; tri_material : (int32, Bigarray.int32_elt, Bigarray.c_layout) Bigarray.Array1.t
  ; floor_vertices : floatarray (* 9 per floor triangle *)
  ; floor_uv : floatarray (* 6 per floor triangle *)
  ; floor_material : (int32, Bigarray.int32_elt, Bigarray.c_layout) Bigarray.Array1.t
  }

type scene (* custom block around the ptx_scene* handle; finalised by the GC *)

external leaf_size : unit -> int = "ptx_ml_leaf_size"
external device_count : unit -> int = "ptx_ml_device_count"
external scene_create_flat : flat -> int -> scene = "ptx_ml_scene_create_stub"
external scene_destroy : scene -> unit = "ptx_ml_scene_destroy_stub"

(* tree depth, nodes, leaves, leaf slots: what main.ml prints after Shape_tree.create (ganesha/bin/main.ml:191-195) *)
external scene_tree_stats : scene -> int * int * int * int = "ptx_ml_scene_tree_stats_stub"

(* Optional: page-lock the image the renders go into (ptx_image_pin): the frame then comes back with one DMA.  A Bigarray's data
   is outside the OCaml heap and does not move; the pin must end (image_unpin, scene_destroy) before the image is collected --
   with_pinned_image below keeps the image reachable for exactly that long. *)
external image_pin : scene -> (float, Bigarray.float64_elt, Bigarray.c_layout) Bigarray.Array1.t -> unit = "ptx_ml_image_pin_stub"
external image_unpin : scene -> unit = "ptx_ml_image_unpin_stub"

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
  = "ptx_ml_render_stub_bytecode" "ptx_ml_render_stub"

external ppm_render_flat
  :  scene
  -> floatarray (* width, height, iterations, max_bounces, photon_count, alpha *)
  -> floatarray (* 11 per light: kind, position xyz, direction xyz, color rgb, power *)
  -> (float, Bigarray.float64_elt, Bigarray.c_layout) Bigarray.Array1.t (* img_sum, W*H*3 *)
  -> (int -> float -> int -> unit) (* iteration, radius, photon map length; img_sum holds the running sum *)
  -> unit
  = "ptx_ml_ppm_render_stub"

module FA = Stdlib.Float.Array

(* the tables are interned structurally: a 150 k-triangle mesh with one material gets one row, not 150 k *)
type tables =
  { tex_rows : float Queue.t
  ; mat_rows : float Queue.t
  ; tex_ids : (Texture.t, int) Hashtbl.Poly.t
  ; mat_ids : (Material.t, int) Hashtbl.Poly.t
  }

let add_texture tb (t : Texture.t) =
  Hashtbl.find_or_add tb.tex_ids t ~default:(fun () ->
    let row =
      match t with
      | Solid (r, g, b) -> [ 0.; 0.; 0.; r; g; b; 0.; 0.; 0. ]
      | Checker { width; height; even = er, eg, eb; odd = or_, og, ob } ->
        [ 1.; Float.of_int width; Float.of_int height; er; eg; eb; or_; og; ob ]
    in
    let idx = Queue.length tb.tex_rows / 9 in
    List.iter row ~f:(Queue.enqueue tb.tex_rows);
    idx)
;;

let add_material tb (m : Material.t) =
  Hashtbl.find_or_add tb.mat_ids m ~default:(fun () ->
    let rec row (m : Material.t) (er, eg, eb) =
      match m with
      | Lambertian t -> [ 0.; Float.of_int (add_texture tb t); 0.; er; eg; eb ]
      | Metal t -> [ 1.; Float.of_int (add_texture tb t); 0.; er; eg; eb ]
      | Dielectric index -> [ 2.; 0.; index; er; eg; eb ]
      | Emitting (inner, emit) -> row inner emit
    in
    let r = row m (0., 0., 0.) in
    let idx = Queue.length tb.mat_rows / 6 in
    List.iter r ~f:(Queue.enqueue tb.mat_rows);
    idx)
;;

let int32_array n ~f =
  let a = Bigarray.Array1.create Bigarray.int32 Bigarray.c_layout n in
  for i = 0 to n - 1 do
    a.{i} <- Int32.of_int_exn (f i)
  done;
  a
;;

let uv6 ((ua, va), (ub, vb), (uc, vc)) = [ ua; va; ub; vb; uc; vc ]

let flatten ~leaf ~(mesh : mesh) ~(floor : floor_triangle list) camera background (spheres : sphere array) =
  let tb =
    { tex_rows = Queue.create ()
    ; mat_rows = Queue.create ()
    ; tex_ids = Hashtbl.Poly.create ()
    ; mat_ids = Hashtbl.Poly.create ()
    }
  in
  let n = Array.length spheres in
  let xs = FA.init n (fun i -> spheres.(i).x)
  and ys = FA.init n (fun i -> spheres.(i).y)
  and zs = FA.init n (fun i -> spheres.(i).z)
  and rs = FA.init n (fun i -> spheres.(i).radius) in
  let sphere_material = int32_array n ~f:(fun i -> add_material tb spheres.(i).material) in
  let n_tri = Bigarray.Array1.dim mesh.faces / 3 in
  if Bigarray.Array1.dim mesh.faces <> 3 * n_tri then invalid_arg "Ptx: mesh.faces holds 3 indices per triangle";
  if (not (Array.is_empty mesh.face_uvs)) && Array.length mesh.face_uvs <> n_tri
  then invalid_arg "Ptx: mesh.face_uvs must be empty or hold one entry per triangle";
  if (not (Array.is_empty mesh.face_materials)) && Array.length mesh.face_materials <> n_tri
  then invalid_arg "Ptx: mesh.face_materials must be empty or hold one entry per triangle";
  let tri_uv =
    let shared = Array.of_list (uv6 mesh.face_uv) in
    if Array.is_empty mesh.face_uvs
    then FA.init (6 * n_tri) (fun k -> shared.(k % 6))
    else (
      let rows = Array.map mesh.face_uvs ~f:(fun t -> Array.of_list (uv6 t)) in
      FA.init (6 * n_tri) (fun k -> rows.(k / 6).(k % 6)))
  in
  let tri_material =
    if Array.is_empty mesh.face_materials
    then (
      let m = if n_tri > 0 then add_material tb mesh.face_material else 0 in
      int32_array n_tri ~f:(fun _ -> m))
    else int32_array n_tri ~f:(fun i -> add_material tb mesh.face_materials.(i))
  in
  let floor = Array.of_list floor in
  let n_floor = Array.length floor in
  let floor_vertices =
    let rows =
      Array.map floor ~f:(fun f ->
        let (ax, ay, az), (bx, by, bz), (cx, cy, cz) = f.vertices in
        [| ax; ay; az; bx; by; bz; cx; cy; cz |])
    in
    FA.init (9 * n_floor) (fun k -> rows.(k / 9).(k % 9))
  in
  let floor_uv =
    let rows = Array.map floor ~f:(fun f -> Array.of_list (uv6 f.uvs)) in
    FA.init (6 * n_floor) (fun k -> rows.(k / 6).(k % 6))
  in
  let floor_material = int32_array n_floor ~f:(fun i -> add_material tb floor.(i).surface) in
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
  ; materials = of_queue tb.mat_rows
  ; textures = of_queue tb.tex_rows
  ; camera = FA.of_list [ camera.lower_left_x; camera.lower_left_y; camera.view_x; camera.view_y ]
  ; background
  ; leaf_kind
  ; length_cutoff
  ; vertex_x = mesh.vx
  ; vertex_y = mesh.vy
  ; vertex_z = mesh.vz
  ; tri_indices = mesh.faces
  ; tri_uv
  ; tri_material
  ; floor_vertices
  ; floor_uv
  ; floor_material
  }
;;

(* Shape_tree.create + uploading the scene: what main.ml does before Render_cmd.run / Ppm.go.  The tree is built over
   [triangles of the mesh, in order] @ [spheres, in order] -- cornell-box's shape list (cornell-box/bin/main.ml:213-218) *)
let scene_create ?(device = 0) ?(leaf = Simd_leaf) ?(mesh = empty_mesh) ?(floor = []) ~camera ~background spheres =
  scene_create_flat (flatten ~leaf ~mesh ~floor camera background spheres) device
;;

(* Bbox of the mesh = union of its triangles' boxes: what Shape_tree.create computes for the root (shape_tree.ml:257-260)
   and what ganesha's Floor and lights are placed from BEFORE the tree exists here (ganesha/bin/main.ml:203-228,267-275) *)
let mesh_bbox (mesh : mesh) =
  let n = Bigarray.Array1.dim mesh.faces in
  if n = 0 then invalid_arg "Ptx.mesh_bbox: empty mesh";
  let lo = [| Float.infinity; Float.infinity; Float.infinity |]
  and hi = [| Float.neg_infinity; Float.neg_infinity; Float.neg_infinity |] in
  for k = 0 to n - 1 do
    let i = Int32.to_int_exn mesh.faces.{k} in
    let p = [| FA.get mesh.vx i; FA.get mesh.vy i; FA.get mesh.vz i |] in
    for a = 0 to 2 do
      lo.(a) <- Float.min lo.(a) p.(a);
      hi.(a) <- Float.max hi.(a) p.(a)
    done
  done;
  (lo.(0), lo.(1), lo.(2)), (hi.(0), hi.(1), hi.(2))
;;

(* Integrator.create ... |> Integrator.render ~update_progress, on [gpus] GPUs of this node *)
let render ?(gpus = 1) scene ~width ~height ~samples_per_pixel ~max_bounces ~image ~update_progress =
  render_flat scene width height samples_per_pixel max_bounces gpus image update_progress
;;

(* [f ()] with [image] pinned; a host that renders many frames into one Bimage (an animation loop around Render_command's run)
   wraps the loop in this.  A pin that cannot be had is not an error: the renders then take the staged copy. *)
let with_pinned_image scene image f =
  (try image_pin scene image with Failure _ -> ());
  Fun.protect f ~finally:(fun () ->
    image_unpin scene;
    ignore (Sys.opaque_identity image))
;;

(* Progressive_photon_map.Make(Scene).go without its prints and its PNG: [img_sum] (W*H*3, the data of a Bimage f64 rgb)
   receives the running sum; [on_iteration] runs on the calling thread after every iteration *)
let ppm_render scene ~width ~height ~iterations ~max_bounces ~photon_count ~alpha ~(lights : light list) ~img_sum ~on_iteration =
  let params =
    FA.of_list
      [ Float.of_int width
      ; Float.of_int height
      ; Float.of_int iterations
      ; Float.of_int max_bounces
      ; Float.of_int photon_count
      ; alpha
      ]
  in
  let rows =
    List.concat_map lights ~f:(function
      | Point { position = px, py, pz; color = r, g, b; power } -> [ 0.; px; py; pz; 0.; 0.; 0.; r; g; b; power ]
      | Spot { position = px, py, pz; direction = dx, dy, dz; color = r, g, b; power } ->
        [ 1.; px; py; pz; dx; dy; dz; r; g; b; power ])
  in
  ppm_render_flat scene params (FA.of_list rows) img_sum (fun iteration radius photon_map_length ->
    on_iteration ~iteration ~radius ~photon_map_length)
;;
