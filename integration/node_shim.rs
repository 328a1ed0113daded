//! UNBUILT SOURCE (no Rust toolchain in the build image; written against Bevy 0.14.2's API from memory): what
//! src/pipeline.rs and src/node.rs of camms205/ray-tracing become when the pass runs on librt_amd.so.
//! `RayTracingPlugin`, `RayTracingGraph`, `RayTracingInfo` and `prepare_meshinfo` (src/ray_tracing.rs:23-169) stay as they
//! are, names and lifecycle included: the plugin still registers `ViewNodeRunner<RayTracingPassNode>` and still creates
//! `RayTracingPipeline` in `finish()` (src/ray_tracing.rs:44-48).  Tested equivalent of this file's logic, call for call:
//! ray-tracing_amd/host/ray_tracing.cpp (`RayTracingPipeline::from_world`, `RayTracingPassNode::run`).
//!
//! Every identifier below resolves to the `rt-amd-sys` crate (integration/rt-amd-sys), to Bevy, or to this file.
//! Three things the maintainer adds outside it:
//!   * `to_repr_c` (section 1) reads the private fields of `Triangle` / `MeshInfo` (src/ray_tracing.rs:51-63), so it lives in
//!     src/ray_tracing.rs next to them;
//!   * assets/shaders/rt_amd_present.wgsl with the six lines of `PRESENT_WGSL` (section 3);
//!   * the camera that opts into `RayTracingGraph` (src/main.rs:59-63) needs nothing new: the pass still writes the view's
//!     out texture, as src/node.rs:63-69 does.
use std::sync::Mutex;

use bevy::{
    core_pipeline::fullscreen_vertex_shader::fullscreen_shader_vertex_state,
    ecs::query::QueryItem,
    prelude::*,
    render::{
        camera::ExtractedCamera,
        render_graph::{NodeRunError, RenderGraphContext, ViewNode},
        render_resource::{
            binding_types::texture_2d, BindGroupEntries, BindGroupLayout, BindGroupLayoutEntries, CachedRenderPipelineId,
            ColorTargetState, ColorWrites, Extent3d, FragmentState, ImageCopyTexture, ImageDataLayout, MultisampleState,
            Origin3d, PipelineCache, PrimitiveState, RenderPassDescriptor, RenderPipelineDescriptor, ShaderStages, Texture,
            TextureAspect, TextureDescriptor, TextureDimension, TextureFormat, TextureSampleType, TextureUsages,
            TextureViewDescriptor,
        },
        renderer::{RenderContext, RenderDevice, RenderQueue},
        view::{ExtractedView, ViewTarget},
    },
};
use rt_amd_sys::*;

use crate::ray_tracing::RayTracingInfo;

// ---- 1. RayTracingInfo -> the C ABI's T1-T4 arrays (goes into src/ray_tracing.rs) ------------------------------------------
// encase lays `MeshInfo` out as std430 (vec3 aligned to 16: 48 B) and `[Vec3; 2]` with a 16-B stride (32 B); the `#[repr(C)]`
// structs of rt-amd-sys spell those paddings out, so this is a field-by-field copy (include/rt_amd.h:35-46).
pub struct SceneArrays {
    pub triangles: Vec<rt_triangle>,
    pub meshes: Vec<rt_mesh_info>,
    pub vertices: Vec<rt_vertex>,
    pub materials: Vec<rt_material>,
}

pub fn to_repr_c(info: &RayTracingInfo) -> SceneArrays {
    SceneArrays {
        triangles: info.triangles.iter().map(|t| rt_triangle { indices: t.indices }).collect(),
        meshes: info
            .meshes
            .iter()
            .map(|m| rt_mesh_info {
                first_tri: m.first_tri,
                tri_count: m.tri_count,
                material: m.material,
                _pad0: 0,
                aabb_min: m.aabb_min.to_array(),
                _pad1: 0.0,
                aabb_max: m.aabb_max.to_array(),
                _pad2: 0.0,
            })
            .collect(),
        vertices: info
            .vertices
            .iter()
            .map(|v| rt_vertex { pos: v[0].to_array(), _pad0: 0.0, norm: v[1].to_array(), _pad1: 0.0 })
            .collect(),
        materials: info.materials.iter().map(|m| rt_material { color: m.color.to_f32_array() }).collect(),
    }
}

// ---- 2. what changed since the last frame ----------------------------------------------------------------------------------
// The reference rebuilds the four vectors every frame (src/ray_tracing.rs:113-168) and main.rs rotates one cube
// (src/main.rs:122-126).  The library wants to hear only about what moved: a topology signature (triangles, mesh ranges,
// materials, counts) decides between rt_upload_scene and an update, one hash per mesh over its vertex run and its AABB finds
// the moved meshes.  FNV-1a over the raw bytes; the host mirror does the same per entity (host/ray_tracing.cpp).
fn fnv1a(seed: u64, bytes: &[u8]) -> u64 {
    let mut h = seed ^ 0xcbf2_9ce4_8422_2325;
    for b in bytes {
        h = (h ^ *b as u64).wrapping_mul(0x0000_0100_0000_01b3);
    }
    h
}
fn bytes_of<T: Copy>(s: &[T]) -> &[u8] {
    // T is one of the padding-explicit #[repr(C)] structs above: every byte is initialised
    unsafe { std::slice::from_raw_parts(s.as_ptr() as *const u8, std::mem::size_of_val(s)) }
}

/// prepare_meshinfo appends every mesh's vertices as one run and rebases its indices by the running vertex count
/// (src/ray_tracing.rs:130,158-162): mesh m's run starts at the smallest index its triangles use and ends where the next
/// mesh's begins.  Returns n_meshes + 1 offsets.
fn vertex_runs(a: &SceneArrays) -> Vec<usize> {
    let mut start = Vec::with_capacity(a.meshes.len() + 1);
    let mut prev = 0usize;
    for m in &a.meshes {
        let tris = &a.triangles[m.first_tri as usize..(m.first_tri + m.tri_count) as usize];
        let lo = tris.iter().flat_map(|t| t.indices).min().map_or(prev, |v| v as usize);
        start.push(lo.max(prev));
        prev = lo.max(prev);
    }
    start.push(a.vertices.len());
    start
}

pub struct Signatures {
    pub topology: u64,
    pub per_mesh: Vec<u64>,
}

pub fn signatures(a: &SceneArrays) -> Signatures {
    let mut topology = fnv1a(a.triangles.len() as u64, bytes_of(&a.triangles));
    topology = fnv1a(topology, bytes_of(&a.materials));
    topology = fnv1a(topology ^ a.vertices.len() as u64, &[]);
    for m in &a.meshes {
        topology = fnv1a(topology, bytes_of(&[m.first_tri, m.tri_count, m.material]));
    }
    let runs = vertex_runs(a);
    let per_mesh = a
        .meshes
        .iter()
        .enumerate()
        .map(|(i, m)| fnv1a(fnv1a(0, bytes_of(&a.vertices[runs[i]..runs[i + 1]])), bytes_of(std::slice::from_ref(m))))
        .collect();
    Signatures { topology, per_mesh }
}

// ---- 3. the pipeline resource: replaces src/pipeline.rs:19-72 --------------------------------------------------------------
/// assets/shaders/rt_amd_present.wgsl.  The frame arrives as the exact Bgra8UnormSrgb bytes the reference's target would hold
/// (rt_present_async, include/rt_amd.h); they are uploaded into a Bgra8UnormSrgb staging texture and copied texel for texel
/// into the view's out texture: sRGB decode on the load and encode on the store of an 8-bit value is the identity.
pub const PRESENT_WGSL: &str = r#"
#import bevy_core_pipeline::fullscreen_vertex_shader::FullscreenVertexOutput
@group(0) @binding(0) var frame: texture_2d<f32>;
@fragment
fn fragment(in: FullscreenVertexOutput) -> @location(0) vec4<f32> {
    return textureLoad(frame, vec2<i32>(in.position.xy), 0);
}
"#;

const FRAMES_IN_FLIGHT: u32 = 2;

struct Pending {
    ticket: u32,
    width: u32,
    height: u32,
}

struct ShimState {
    topology: u64,
    per_mesh: Vec<u64>,
    frame: u64,
    /// one host frame per set in flight; rt_present_async copies into it, rt_present_wait makes it readable
    host: [Vec<u8>; FRAMES_IN_FLIGHT as usize],
    pending: [Option<Pending>; FRAMES_IN_FLIGHT as usize],
    staging: Option<(Texture, u32, u32)>,
}

#[derive(Resource)]
pub struct RayTracingPipeline {
    ctx: *mut rt_ctx,
    state: Mutex<ShimState>,
    /// layout + cached id of the six-line presentation pipeline (same fields as the reference's struct, src/pipeline.rs:20-23)
    pub layout: BindGroupLayout,
    pub pipeline_id: CachedRenderPipelineId,
}
// The context is not re-entrant (include/rt_amd.h:10-12); Bevy calls `run` from its one render thread, and the state that
// changes per frame sits behind the mutex.
unsafe impl Send for RayTracingPipeline {}
unsafe impl Sync for RayTracingPipeline {}

impl FromWorld for RayTracingPipeline {
    fn from_world(world: &mut World) -> Self {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { rt_create(0, &mut ctx) };
        assert_eq!(rc, RT_OK, "rt_create: no gfx950 device (the library has no CPU fallback)");
        // the node submits frame after frame and never waits (src/node.rs:29-79): two sets of frame buffers, so frame k + 1
        // renders while frame k's bytes travel to the host
        let rc = unsafe { rt_set_frames_in_flight(ctx, FRAMES_IN_FLIGHT) };
        assert_eq!(rc, RT_OK, "rt_set_frames_in_flight");

        let render_device = world.resource::<RenderDevice>();
        let layout = render_device.create_bind_group_layout(
            "rt_amd_present_layout",
            &BindGroupLayoutEntries::single(ShaderStages::FRAGMENT, texture_2d(TextureSampleType::Float { filterable: false })),
        );
        let shader = world.resource::<AssetServer>().load("shaders/rt_amd_present.wgsl");
        let pipeline_id = world.resource::<PipelineCache>().queue_render_pipeline(RenderPipelineDescriptor {
            label: Some("rt_amd_present_pipeline".into()),
            layout: vec![layout.clone()],
            push_constant_ranges: vec![],
            vertex: fullscreen_shader_vertex_state(),
            primitive: PrimitiveState::default(),
            depth_stencil: None,
            multisample: MultisampleState::default(),
            fragment: Some(FragmentState {
                shader,
                shader_defs: vec![],
                entry_point: "fragment".into(),
                targets: vec![Some(ColorTargetState {
                    format: TextureFormat::Bgra8UnormSrgb, // the reference's target format, src/pipeline.rs:57
                    blend: None,
                    write_mask: ColorWrites::ALL,
                })],
            }),
        });
        Self {
            ctx,
            state: Mutex::new(ShimState {
                topology: 0,
                per_mesh: Vec::new(),
                frame: 0,
                host: [Vec::new(), Vec::new()],
                pending: [None, None],
                staging: None,
            }),
            layout,
            pipeline_id,
        }
    }
}

impl Drop for RayTracingPipeline {
    fn drop(&mut self) {
        unsafe {
            rt_synchronize(self.ctx); // the host frames below are targets of copies in flight
            rt_destroy(self.ctx);
        }
    }
}

fn check(ctx: *mut rt_ctx, rc: i32, what: &str) {
    if rc != RT_OK {
        let msg = unsafe { std::ffi::CStr::from_ptr(rt_last_error(ctx)) }.to_string_lossy().into_owned();
        panic!("{what}: status {rc}: {msg}"); // the reference unwrap()s its way through run() the same way (src/node.rs:38-49)
    }
}

/// Scene to the device: a rebuild when the topology changed, else only what moved.
fn sync_scene(ctx: *mut rt_ctx, st: &mut ShimState, a: &SceneArrays) {
    let sig = signatures(a);
    if sig.topology != st.topology || sig.per_mesh.len() != st.per_mesh.len() {
        let rc = unsafe {
            rt_upload_scene(
                ctx,
                a.triangles.as_ptr(), a.triangles.len(),
                a.meshes.as_ptr(), a.meshes.len(),
                a.vertices.as_ptr(), a.vertices.len(),
                a.materials.as_ptr(), a.materials.len(),
            )
        };
        check(ctx, rc, "rt_upload_scene");
    } else {
        let moved: Vec<usize> = (0..sig.per_mesh.len()).filter(|&i| sig.per_mesh[i] != st.per_mesh[i]).collect();
        if let (Some(&first), Some(&last)) = (moved.first(), moved.last()) {
            let n = last - first + 1;
            let rc = if 2 * n <= a.meshes.len() {
                // a Changed<GlobalTransform> run (the reference's rotating cube, src/main.rs:122-126): only these meshes'
                // T2 entries and their vertex run cross PCIe, only their leaves are re-baked, only their ancestors refitted
                let runs = vertex_runs(a);
                let (v0, v1) = (runs[first], runs[last + 1]);
                unsafe {
                    rt_update_vertex_range(ctx, a.vertices[v0..v1].as_ptr(), v0, v1 - v0, a.meshes[first..=last].as_ptr(), first, n)
                }
            } else {
                unsafe { rt_update_vertices(ctx, a.vertices.as_ptr(), a.vertices.len(), a.meshes.as_ptr(), a.meshes.len()) }
            };
            if rc == RT_ERR_SCENE {
                // e.g. a vertex became non-finite, or the built tree had dropped such a triangle: rebuild
                let rc = unsafe {
                    rt_upload_scene(
                        ctx,
                        a.triangles.as_ptr(), a.triangles.len(),
                        a.meshes.as_ptr(), a.meshes.len(),
                        a.vertices.as_ptr(), a.vertices.len(),
                        a.materials.as_ptr(), a.materials.len(),
                    )
                };
                check(ctx, rc, "rt_upload_scene (after a refused update)");
            } else {
                check(ctx, rc, "rt_update_vertex_range / rt_update_vertices");
            }
        }
    }
    st.topology = sig.topology;
    st.per_mesh = sig.per_mesh;
}

/// Bytes of a finished frame into the staging texture (created per size), ready for the presentation pass.
fn write_to_target<'a>(device: &RenderDevice, queue: &RenderQueue, st: &'a mut ShimState, slot: usize, width: u32, height: u32) -> &'a Texture {
    let fresh = !matches!(&st.staging, Some((_, w, h)) if *w == width && *h == height);
    if fresh {
        let tex = device.create_texture(&TextureDescriptor {
            label: Some("rt_amd_frame"),
            size: Extent3d { width, height, depth_or_array_layers: 1 },
            mip_level_count: 1,
            sample_count: 1,
            dimension: TextureDimension::D2,
            format: TextureFormat::Bgra8UnormSrgb,
            usage: TextureUsages::COPY_DST | TextureUsages::TEXTURE_BINDING,
            view_formats: &[],
        });
        st.staging = Some((tex, width, height));
    }
    let tex = &st.staging.as_ref().unwrap().0;
    queue.write_texture(
        ImageCopyTexture { texture: tex, mip_level: 0, origin: Origin3d::ZERO, aspect: TextureAspect::All },
        &st.host[slot],
        ImageDataLayout { offset: 0, bytes_per_row: Some(4 * width), rows_per_image: Some(height) },
        Extent3d { width, height, depth_or_array_layers: 1 },
    );
    tex
}

// ---- 4. the node: replaces src/node.rs:18-80 -------------------------------------------------------------------------------
#[derive(Default)]
pub struct RayTracingPassNode;

impl ViewNode for RayTracingPassNode {
    // the reference's query minus what only its shader read (prepass textures, the view uniform's dynamic offset,
    // src/node.rs:22-27), plus the extracted view the `View` uniform is made from
    type ViewQuery = (&'static ExtractedCamera, &'static ExtractedView, &'static ViewTarget);

    fn run(
        &self,
        _graph: &mut RenderGraphContext,
        render_context: &mut RenderContext,
        (camera, view, target): QueryItem<Self::ViewQuery>,
        world: &World,
    ) -> Result<(), NodeRunError> {
        let info = world.resource::<RayTracingInfo>();
        let pipe = world.resource::<RayTracingPipeline>();
        let device = world.resource::<RenderDevice>();
        let queue = world.resource::<RenderQueue>();
        let mut st = pipe.state.lock().unwrap();
        let ctx = pipe.ctx;

        // bind group 1 (src/node.rs:42-50): sent when it changed, not every frame
        sync_scene(ctx, &mut st, &to_repr_c(info));

        // bind group 0, the `View` uniform (ray_tracing.wgsl:5,176-177): world_from_clip and world_position as
        // bevy_render's prepare_view_uniforms derives them, plus the camera's viewport (src/node.rs:71-73)
        let view_from_clip = view.clip_from_view.inverse();
        let world_from_clip = match view.clip_from_world {
            Some(clip_from_world) => clip_from_world.inverse(),
            None => view.world_from_view.compute_matrix() * view_from_clip,
        };
        let m = world_from_clip.to_cols_array(); // column-major, as rt_set_view wants it
        let p = view.world_from_view.translation().to_array();
        let size = target.main_texture().size();
        let (width, height) = (size.width, size.height);
        let vp = camera
            .viewport
            .as_ref()
            .map(|v| [v.physical_position.x, v.physical_position.y, v.physical_size.x, v.physical_size.y]);
        let rc = unsafe { rt_set_view(ctx, m.as_ptr(), p.as_ptr(), width, height, vp.as_ref().map_or(std::ptr::null(), |v| v.as_ptr())) };
        check(ctx, rc, "rt_set_view");

        // draw(0..3, 0..1) (src/node.rs:77)
        let params = rt_render_params {
            struct_size: std::mem::size_of::<rt_render_params>() as u32,
            frame_count: info.count, // the `count` uniform (src/ray_tracing.rs:86-87)
            ..Default::default()
        };
        match unsafe { rt_render(ctx, &params) } {
            RT_OK => {}
            RT_NOT_READY => return Ok(()), // src/node.rs:52-56: skip the frame silently
            rc => check(ctx, rc, "rt_render"),
        }

        // The colour attachment (src/node.rs:63-69).  Encode + copy of THIS frame go onto its own stream and return at once;
        // what is presented now is the frame submitted one call ago, whose bytes have had a frame's time to arrive -- a
        // blocking rt_read_bgra8_srgb here would drain the GPU every frame (1.35 ms against 0.63 ms per 1080p frame).
        let slot = (st.frame % FRAMES_IN_FLIGHT as u64) as usize;
        let prev = ((st.frame + FRAMES_IN_FLIGHT as u64 - 1) % FRAMES_IN_FLIGHT as u64) as usize;
        st.frame += 1;
        if let Some(old) = st.pending[slot].take() {
            // this slot's host frame is about to be overwritten: its copy must have ended (it has, a frame ago)
            check(ctx, unsafe { rt_present_wait(ctx, old.ticket) }, "rt_present_wait");
        }
        let n_bytes = (width as usize) * (height as usize) * 4;
        st.host[slot].resize(n_bytes, 0);
        let mut ticket = 0u32;
        let rc = unsafe { rt_present_async(ctx, st.host[slot].as_mut_ptr(), n_bytes, &mut ticket) };
        check(ctx, rc, "rt_present_async");
        st.pending[slot] = Some(Pending { ticket, width, height });

        let Some(done) = st.pending[prev].take() else {
            return Ok(()); // the first frame: nothing has arrived yet
        };
        check(ctx, unsafe { rt_present_wait(ctx, done.ticket) }, "rt_present_wait");
        if (done.width, done.height) != (width, height) {
            return Ok(()); // the window was resized in between: that frame no longer fits
        }
        let Some(pipeline) = world.resource::<PipelineCache>().get_render_pipeline(pipe.pipeline_id) else {
            return Ok(()); // the presentation pipeline is still compiling (same silent skip, src/node.rs:52-56)
        };
        let frame_view = write_to_target(device, queue, &mut st, prev, width, height).create_view(&TextureViewDescriptor::default());
        let bind_group = device.create_bind_group("rt_amd_present_bind_group", &pipe.layout, &BindGroupEntries::single(&frame_view));
        let mut render_pass = render_context.begin_tracked_render_pass(RenderPassDescriptor {
            label: Some("ray_tracing_render_pass"),
            color_attachments: &[Some(target.out_texture_color_attachment(None))],
            depth_stencil_attachment: None,
            timestamp_writes: None,
            occlusion_query_set: None,
        });
        // no set_camera_viewport: the library already left pixels outside the viewport cleared (rt_set_view)
        render_pass.set_render_pipeline(pipeline);
        render_pass.set_bind_group(0, &bind_group, &[]);
        render_pass.draw(0..3, 0..1);
        Ok(())
    }
}
