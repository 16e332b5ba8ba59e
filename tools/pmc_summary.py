"""summarise rocprofv3 --pmc counter_collection.csv: mean counter value per
kernel (our kernels only).
usage: pmc_summary.py <csv> "<COUNTER> [<COUNTER> ...]" [json_out] [kernel_trace.csv]
With a kernel trace of the same run the mean launch duration [ns] of every
kernel is added under "_duration_ns" (what the SQ cycle counters of a
profiled pass are divided by: the counters and the duration come from the SAME
launches)."""
import collections
import csv
import json
import re
import sys

path, counter = sys.argv[1], sys.argv[2]


def short_name(name):
    m = re.search(r'nice_map_fused_kernel<(\d+), (\d+), (\w+), (\w+)(?:, \w+)?>', name)
    if m:
        return (f'nice_map_fused<stage={m.group(1)},NT={m.group(2)},'
                f'dp={m.group(3)},dw={m.group(4)}>')
    m = re.search(r'nice_bwd_fused_kernel<(\d+), (\d+), (\w+), (\w+)'
                  r'(?:, \d+)?>', name)
    if m:
        return (f'nice_bwd_fused<stage={m.group(1)},NT={m.group(2)},'
                f'dp={m.group(3)},dw={m.group(4)}>')
    m = re.search(r'nice_fwd_kernel<(\d+), (\d+)(?:, \d+)?>', name)
    if m:
        return f'nice_fwd<stage={m.group(1)},NT={m.group(2)}>'
    m = re.search(r'vox_points_bwd_kernel<(\w+)>', name)
    if m:
        return f'vox_points_bwd<dw={m.group(1)}>'
    m = re.search(r'coslam_bwd_kernel<(\w+), (\w+)>', name)
    if m:
        return f'coslam_bwd<dp={m.group(1)},dg={m.group(2)}>'
    for k in ('nice_fwd_roles_finish_kernel', 'nice_fwd_roles_kernel',
              'nice_bwd_roles_kernel',
              'nice_map_coarse_finish_kernel', 'nice_map_coarse_kernel',
              'nice_map_finish_kernel',
              'nice_bwd_coarse_kernel', 'nice_bwd_finish_kernel',
              'coarse_rep_reduce_kernel', 'adam_cells_kernel', 'coslam_fwd_kernel',
              'hash_chunk_scatter_kernel', 'hash_chunk_scatter_runs_kernel',
              'coslam_reduce_kernel',
              'coslam_loss_grad_kernel', 'adam_dense_kernel',
              'reduce_partials_kernel', 'hashgrid_kernel',
              'vox_points_fwd_kernel', 'vox_points_bwd_kernel', 'vox_dw_kernel',
              'vox_dw_reduce_kernel', 'vox_sample_kernel',
              'svo_intersect_kernel', 'vox_hit_sort_kernel',
              'vox_compact_kernel', 'vox_render_fwd_kernel',
              'vox_render_bwd_kernel', 'vox_ray_grad_kernel',
              'gs_render_fwd_kernel', 'gs_render_bwd_kernel',
              'gs_preprocess_kernel', 'knn_search_kernel',
              'point_color_bwd_w_kernel', 'vox_hits_kernel',
              'point_color_bwd_kernel', 'point_color_fwd_kernel',
              'pc_dw_reduce_kernel', 'pc_dw_kernel', 'point_geo_bwd_kernel',
              'point_geo_fwd_kernel', 'point_map_loss_kernel',
              'gs_blend_fwd_kernel', 'gs_blend_bwd_kernel',
              'gs_key_reduce_kernel', 'gs_pack_kernel',
              'gs_prepare_fwd_kernel', 'gs_prepare_bwd_kernel',
              'gs_loss_stats_kernel', 'gs_loss_grad_kernel',
              'frustum_select_kernel', 'tv_points_kernel', 'tv_loss_kernel',
              'coslam_extra_stage_kernel', 'coslam_map_rows_kernel',
              'frustum_depth_kernel'):
        if k in name:
            return k
    return None


counters = counter.split()
acc = {c: collections.defaultdict(list) for c in counters}
for r in csv.DictReader(open(path)):
    c = r.get('Counter_Name')
    if c not in acc:
        continue
    s = short_name(r['Kernel_Name'])
    if s is not None:
        acc[c][s].append(float(r['Counter_Value']))
res = {}
for c in counters:
    out = {}
    for k, v in sorted(acc[c].items()):
        out[k] = {'launches': len(v), 'mean': sum(v) / len(v), 'max': max(v)}
        print(f'{k:48s} launches={len(v):5d} mean_{c}={sum(v)/len(v):14.1f} '
              f'max={max(v):14.1f}')
    res[c] = out
if len(sys.argv) > 4 and sys.argv[4]:
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[4])):
        s = short_name(r['Kernel_Name'])
        if s is not None:
            dur[s].append(float(r['End_Timestamp']) -
                          float(r['Start_Timestamp']))
    res['_duration_ns_' + counters[0]] = {
        k: {'launches': len(v), 'mean': sum(v) / len(v)}
        for k, v in sorted(dur.items())}
if len(sys.argv) > 3:
    json.dump(res, open(sys.argv[3], 'w'), indent=1)
