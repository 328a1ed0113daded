//! UNBUILT SOURCE: what src/pipeline.rs and src/node.rs of camms205/ray-tracing become when the pass
//! runs on librt_amd.so.  RayTracingPlugin, RayTracingGraph, RayTracingInfo and prepare_meshinfo
//! (src/ray_tracing.rs) stay as they are.  Tested equivalent: ray-tracing_amd/host/ray_tracing.cpp.
use bevy::{ecs::query::QueryItem, prelude::*, render::{camera::ExtractedCamera, render_graph::{NodeRunError, RenderGraphContext, ViewNode},
    renderer::RenderContext, view::{ExtractedView, ViewTarget}}};
use rt_amd_sys::*;
use crate::ray_tracing::RayTracingInfo;

#[derive(Resource)]
pub struct RayTracingPipeline { ctx: *mut rt_ctx, uploaded: std::sync::atomic::AtomicU64, topology: std::sync::atomic::AtomicU64 }
unsafe impl Send for RayTracingPipeline {}
unsafe impl Sync for RayTracingPipeline {}

impl FromWorld for RayTracingPipeline {           // replaces src/pipeline.rs:25-72
    fn from_world(_: &mut World) -> Self {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { rt_create(0, &mut ctx) };
        assert_eq!(rc, RT_OK, "rt_create: no gfx950 device (the library has no CPU fallback)");
        Self { ctx, uploaded: 0.into(), topology: 0.into() }
    }
}
impl Drop for RayTracingPipeline { fn drop(&mut self) { unsafe { rt_destroy(self.ctx); } } }

#[derive(Default)]
pub struct RayTracingPassNode;
impl ViewNode for RayTracingPassNode {             // replaces src/node.rs:21-79
    type ViewQuery = (&'static ExtractedCamera, &'static ExtractedView, &'static ViewTarget);
    fn run(&self, _g: &mut RenderGraphContext, render_context: &mut RenderContext,
           (camera, view, target): QueryItem<Self::ViewQuery>, world: &World) -> Result<(), NodeRunError> {
        use std::sync::atomic::Ordering::Relaxed;
        let info = world.resource::<RayTracingInfo>();
        let pipe = world.resource::<RayTracingPipeline>();
        let (tris, meshes, verts, mats) = to_repr_c(info);          // std430 layouts of T1-T4
        let (sig, topo) = signatures(&tris, &meshes, &verts, &mats);
        unsafe {
            if sig != pipe.uploaded.load(Relaxed) {
                let refit = topo == pipe.topology.load(Relaxed)
                    && rt_update_vertices(pipe.ctx, verts.as_ptr(), verts.len(), meshes.as_ptr(), meshes.len()) == RT_OK;
                if !refit {
                    let rc = rt_upload_scene(pipe.ctx, tris.as_ptr(), tris.len(), meshes.as_ptr(), meshes.len(),
                                             verts.as_ptr(), verts.len(), mats.as_ptr(), mats.len());
                    assert_eq!(rc, RT_OK);
                    pipe.topology.store(topo, Relaxed);
                }
                pipe.uploaded.store(sig, Relaxed);
            }
            let m = view.world_from_clip.unwrap_or(view.world_from_view.compute_matrix() * view.clip_from_view.inverse()).to_cols_array();
            let p = view.world_from_view.translation().to_array();
            let size = target.main_texture().size();
            let vp = camera.viewport.as_ref().map(|v| [v.physical_position.x, v.physical_position.y, v.physical_size.x, v.physical_size.y]);
            rt_set_view(pipe.ctx, m.as_ptr(), p.as_ptr(), size.width, size.height,
                        vp.as_ref().map_or(std::ptr::null(), |v| v.as_ptr()));
            let params = rt_render_params { struct_size: 48, frame_count: info.count, ..Default::default() };
            match rt_render(pipe.ctx, &params) {
                RT_OK => {}
                RT_NOT_READY => return Ok(()),             // src/node.rs:52-56: skip the frame silently
                _ => panic!("rt_render failed"),
            }
            let mut bgra = vec![0u8; (size.width * size.height * 4) as usize];
            rt_read_bgra8_srgb(pipe.ctx, bgra.as_mut_ptr(), bgra.len());
            write_to_target(render_context, target, &bgra);         // queue.write_texture into out_texture()
        }
        Ok(())
    }
}
