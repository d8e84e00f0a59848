"""Result files of a design run in the reference's wire formats.

Counterpart of ``utils/stats_inputs_outputs.py``: ``sort_trajectory`` / ``generate_trajectory_csv`` (:304-335),
``generate_multifasta`` (:337-358), ``generate_best_fasta`` (:360-382), ``sort_and_filter_simulation_data`` (:384-430, the
single-chain branch), ``sort_and_filter_simulation_data_alternative`` (:422-460, with ``get_alt_mcc``), ``generate_csv_from_data`` (:463-478), ``check_if_design_solved`` / ``write_best_str_file``
(:480-524), ``generate_simulation_stats_text`` / ``write_stats_to_file`` (:526-577), ``get_outname`` (:636-669) and of
``round_floats`` (``utils/sequence_utils.py:1254-1274``).  A record is ``vars(ScoreSeq)``: the CSV header is its key order.

Not written: ``_replicas.csv`` (a pandas pivot), the PNG plot, ``_random.csv``.
"""
import csv
import io
import time


def round_floats(obj):
    if isinstance(obj, float):
        return round(obj, 3)
    if isinstance(obj, dict):
        return {k: round_floats(v) for k, v in obj.items()}
    if isinstance(obj, list):                      # the reference passes lists of records through unchanged ...
        return obj
    return obj


def _round_records(records):
    # ... because round_floats(list) returns the list itself: floats inside the records are NOT rounded by
    # sort_trajectory / sort_and_filter_simulation_data (reference quirk, kept)
    return round_floats(records)


def sort_trajectory(simulation_data):
    return sorted(_round_records(simulation_data), key=lambda d: (d['sim_step'], d['replica_num']))


def _csv_text(rows):
    buf = io.StringIO(newline='')
    w = csv.DictWriter(buf, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in rows:
        w.writerow(r)
    return buf.getvalue()


def trajectory_csv_text(sorted_data):
    return _csv_text(sorted_data)


def multifasta_text(sorted_data, infile, now):
    return "".join(f">{infile}|{now}|{it['replica_num']}|{it['sim_step']}|{it['scoring_function']}\n{it['sequence']}\n"
                   for it in sorted_data)


def best_fasta_text(simulation_data, infile, now, num_results=10):
    top = sorted(_round_records(simulation_data), key=lambda d: (-d['scoring_function']), reverse=True)[:num_results]
    return "".join(f">{infile}|{now}|{s['replica_num']}|{s['sim_step']}|{s['scoring_function']}\n{s['sequence']}\n" for s in top)


def sort_and_filter(simulation_data, num_results=10, oligo_state="none", subopt="off", sec_struct=None):
    """Unique sequences (last occurrence wins), best first (reference ``sort_and_filter_simulation_data``,
    ``utils/stats_inputs_outputs.py:384-419``): lowest 1-MCC, then by branch -- ``-oa on``: scoring function, Ed-Epf, Epf;
    ``-d on``: oligomer fraction (LARGEST first for two different strands, smallest first for two equal ones), Ed-Epf, Epf;
    ``-nd on``: Ed-Epf, then the LARGEST Esubopt-Epf; plain: Ed-Epf, Epf, scoring function."""
    uniq = list({item['sequence']: item for item in simulation_data}.values())
    if oligo_state == "avoid":
        key = lambda d: (-d['mcc'], -d['scoring_function'], -d['edesired_minus_Epf'], -d['Epf'])
    elif oligo_state == "homodimer":
        a, b = (sec_struct or "&").split("&")[:2]
        sign = 1.0 if a != b else -1.0
        key = lambda d: (-d['mcc'], sign * d['oligo_fraction'], -d['edesired_minus_Epf'], -d['Epf'])
    elif subopt != "off":
        key = lambda d: (-d['mcc'], -d['edesired_minus_Epf'], d['esubopt_minus_Epf'])
    else:
        key = lambda d: (-d['mcc'], -d['edesired_minus_Epf'], -d['Epf'], -d['scoring_function'])
    return sorted(_round_records(uniq), key=key, reverse=True)[:num_results]


def get_alt_mcc(simulation_data, alt_sec_structs, engine):
    """1-MCC of every alternative target against the sequence's k-th sub-optimal structure (k = 1 .. #alternatives), added to
    the records as ``mcc_k`` / ``alt_struct_k`` -- reference ``get_alt_mcc`` (``utils/sequence_utils.py:766-793``) on top of
    ``get_first_suboptimal_structure_and_energy`` (``utils/energy_scores.py:453-488``: entry k of the energy-sorted subopt list
    found within at most 49 kcal/mol of the ground state, else all dots).  The ranked structures of all records come from
    one ``engine.subopt_structs`` call per sequence length (GPU; no CPU fallback)."""
    from .sim_score import SimScore
    n_alt = len(alt_sec_structs)
    if not simulation_data or not n_alt:
        return simulation_data
    if n_alt > 7:
        raise ValueError("at most 7 alternative structures (the engine ranks up to 8 structures per sequence)")
    by_len = {}
    for i, d in enumerate(simulation_data):
        if "&" in d["sequence"]:
            raise ValueError("ranked sub-optimal structures are single-strand only")
        by_len.setdefault(len(d["sequence"]), []).append(i)
    for L, idx in by_len.items():
        E, ss = engine.subopt_structs([simulation_data[i]["sequence"] for i in idx], n_alt + 1)
        for row, i in enumerate(idx):
            for k in range(1, n_alt + 1):
                ok = E[row, k] < 10000000 and int(E[row, k]) - int(E[row, 0]) <= 4900
                sub = ss[row][k] if ok else "." * L
                sc = SimScore(alt_sec_structs[k - 1].replace("&", "Ee"), sub.replace("&", "Ee"))
                sc.find_basepairs()
                sc.cofusion_matrix()
                simulation_data[i]["mcc_" + str(k)] = 1 - sc.mcc()
                simulation_data[i]["alt_struct_" + str(k)] = sub
    return simulation_data


def sort_and_filter_alternative(simulation_data, alt_sec_structs, engine, num_results=10):
    """Final ranking of an alternative-structure design (reference ``sort_and_filter_simulation_data_alternative``,
    ``utils/stats_inputs_outputs.py:422-460``): unique sequences sorted as in ``sort_and_filter``, ``get_alt_mcc`` on all
    of them, then best first by 1-MCC of the target and of every alternative, 1-MCC again, scoring function.  Returns
    (top records, all records with the added columns -- the reference rewrites ``_traj.csv`` from the latter)."""
    uniq = list({item['sequence']: item for item in simulation_data}.values())
    first = sorted(_round_records(uniq), key=lambda d: (-d['mcc'], -d['edesired_minus_Epf'], -d['Epf'], -d['scoring_function']),
                   reverse=True)
    aug = get_alt_mcc(first, alt_sec_structs, engine)
    cols = ['mcc'] + ["mcc_" + str(k + 1) for k in range(len(alt_sec_structs))]
    res = sorted(_round_records(aug), key=lambda d: tuple([-d[c] for c in cols] + [-d['mcc'], -d['scoring_function']]), reverse=True)
    return res[:num_results], aug


def results_csv_text(sorted_results):
    return _csv_text(sorted_results)


def check_if_design_solved(sorted_results, input_name):
    correct_count = sum(r['mcc'] == 0.0 for r in sorted_results[:10])
    correct_bool = correct_count > 0
    txt = f">{input_name},{correct_bool},{correct_count},{sorted_results[0]['sequence']},{sorted_results[0]['mfe_ss']}"
    return txt, correct_bool


def stats_text(stats, sorted_results, correct_bool, finish_time, outname, timlim):
    """``stats``: object with step, global_step, acc_mc_step, acc_mc_better_e, rej_mc_step, acc_re_step, rej_re_step."""
    sum_mc = stats.acc_mc_step + stats.rej_mc_step
    acc_perc = round(stats.acc_mc_step / sum_mc, 3) if sum_mc else 0
    sum_mc_metro = sum_mc - stats.acc_mc_better_e
    acc_metro = stats.acc_mc_step - stats.acc_mc_better_e
    sum_replica_att = stats.acc_re_step + stats.rej_re_step if stats.acc_re_step + stats.rej_re_step else 1
    formatted_time = time.strftime("%H:%M:%S", time.gmtime(finish_time))
    best = "\nDesign solved succesfully!\n\nBest solution:\n" if correct_bool else "\nDesign not solved!\n\nTarget structure:\n"
    best += f"{sorted_results[0]['sequence']}\nMFE Secondary Structure: \n{sorted_results[0]['mfe_ss']}\nPartition Function Energy: {round(sorted_results[0]['Epf'], 3)}\
                        \n1-MCC: {round(sorted_results[0]['mcc'], 3)}\n"
    return f"\n>{outname} \ntime={timlim}s\n\nAcc_ratio={acc_perc}, Iterations={stats.step}, Accepted={stats.acc_mc_step}/{sum_mc}, Rejected={stats.rej_mc_step}/{sum_mc}\n" \
           f"Accepted Metropolis={acc_metro}/{sum_mc_metro}, Rejected Metropolis={sum_mc_metro - acc_metro}/{sum_mc_metro}\n" \
           f"Replica exchange attempts: {stats.global_step}\nReplica swaps attempts: {sum_replica_att}\nReplica swaps accepted: {stats.acc_re_step}\n" \
           f"Replica swaps rejected: {stats.rej_re_step}\nReplica exchange acc_ratio: {round(stats.acc_re_step / sum_replica_att, 3)}\n{best}\n\n" \
           f"Simulation time: {formatted_time}\n"


def get_outname(infile, replicas, RE_attempt, timlim, pks, acgu_percentages, T_min, T_max, param, scoring_f, oligo, dimer,
                point_mutations):
    sf = "_".join([f"{func}_{weight}" for func, weight in scoring_f])
    return (infile.split(".")[0] + '_R' + str(replicas) + "_e" + str(RE_attempt) + "_t" + str(timlim) + "_pk" + str(pks) +
            "_ACGU" + str(acgu_percentages) + "_Tmin" + str(T_min) + "_Tmax" + str(T_max) + "_p" + str(param) + "_SF" + sf +
            "_O" + str(oligo) + "_D" + str(dimer) + "_PM" + str(point_mutations))


def write_all(simulation_data, input_name, infile, outname, stats, finish_time, timlim, now, num_results=10, directory=".",
              alt_sec_structs=None, engine=None, oligo_state="none", subopt="off", sec_struct=None):
    """Write _traj.csv, _multifasta.fas, _best_fasta.fas, _results.csv, _best_str and _stats (reference
    parse_and_output_results, :593-633).  With alternative structures the ranking is sort_and_filter_alternative and
    _traj.csv is rewritten from its augmented records, as the reference does.  Returns (sorted_results, solved)."""
    import os
    base = os.path.join(directory, outname)
    traj = sort_trajectory(simulation_data)
    with open(base + '_traj.csv', 'w', newline='', encoding='utf-8') as fh:
        fh.write(trajectory_csv_text(traj))
    with open(base + '_multifasta.fas', 'w', encoding='utf-8') as fh:
        fh.write(multifasta_text(traj, infile, now))
    with open(base + '_best_fasta.fas', 'w', encoding='utf-8') as fh:
        fh.write(best_fasta_text(simulation_data, infile, now, num_results))
    if alt_sec_structs:
        res, aug = sort_and_filter_alternative(simulation_data, alt_sec_structs, engine, num_results)
        with open(base + '_traj.csv', 'w', newline='', encoding='utf-8') as fh:
            fh.write(trajectory_csv_text(aug))
    else:
        res = sort_and_filter(simulation_data, num_results, oligo_state=oligo_state, subopt=subopt, sec_struct=sec_struct)
    with open(base + '_results.csv', 'w', newline='', encoding='utf-8') as fh:
        fh.write(results_csv_text(res))
    txt, ok = check_if_design_solved(res[:10], input_name)
    with open(base + '_best_str', 'w', newline='', encoding='utf-8') as fh:
        fh.write(txt)
    with open(base + '_stats', 'w', newline='\n', encoding='utf-8') as fh:
        fh.write(stats_text(stats, res, ok, finish_time, outname, timlim))
    return res, ok
