"""Quarterly-results JSON -> "complete context" text chunks (host-side Python).

Restates `create_complete_context_chunks(quarterly_data, period)` and the driver
loop of the reference's ingest script ("chunking_storing (1).py":31-332 and
:335-374).  This is string/float formatting with no numeric hot loop, so it stays
on the host; its output is pinned byte for byte by the reference's own
FinRag_knowledge_graph/chunks.json (tests/golden/chunks_golden.json).

Behaviour kept on purpose:
  * files are classified by content, not by name (reportType / payload keys), in
    the reference's precedence order (:51-64);
  * the period label selects the JSON month keys by substring ("Q1" + "2024" ->
    june2023 vs june2022, ...) (:78-89);
  * the ratios chunk is emitted only if its text exceeds 100 characters, and its
    primary_value falls back to 0 when no EPS block was seen (:231-238) -- the
    reference reads stale locals there; `_Carry` models that explicitly;
  * an unrecognised segmental structure ends the function early, keeping the
    chunks built so far (:271-275).
"""
from __future__ import annotations

import glob
import json
import os
from dataclasses import dataclass, field

QUARTERS = ("q1_2023", "q2_2023", "q3_2023", "q4_2023")
PERIOD_OF = {"q1_2023": "Q1_FY2024", "q2_2023": "Q2_FY2024", "q3_2023": "Q3_FY2024",
             "q4_2023": "Q4_FY2024"}

# quarter tag -> (month key, year if "2024" in the period label else year - 1 ... )
_MONTHS = (("Q1", "june", 2023), ("Q2", "september", 2023), ("Q3", "december", 2023),
           ("Q4", "march", 2024))
_SEGMENTS = (("retailBanking", "Retail Banking"), ("wholesaleBanking", "Wholesale Banking"),
             ("treasury", "Treasury"), ("lifeInsurance", "Life Insurance"), ("others", "Others"))


def period_keys(period: str):
    """("june2023", "june2022") for "Q1_FY2024"; one year earlier for other labels."""
    for tag, month, year in _MONTHS:
        if tag in period:
            shift = 0 if "2024" in period else 1
            return f"{month}{year - shift}", f"{month}{year - shift - 1}"
    return None, None


def classify(quarterly_data):
    """Pick the consolidated financial / segmental / balance-sheet documents."""
    fin = seg = bal = None
    for doc in quarterly_data:
        kind = doc.get("reportType")
        if kind == "CONSOLIDATED FINANCIAL RESULTS":
            fin = doc
        elif kind == "CONSOLIDATED SEGMENTAL RESULTS":
            seg = doc
        elif "consolidatedSegmentalResults" in doc:
            seg = doc
        elif "consolidatedBalanceSheet" in doc:
            bal = doc
        elif "segmentalResults" in doc and not seg:
            seg = doc
    return fin, seg, bal


def _pct(num, den):
    return (num / den * 100) if den else 0


def _growth(cur, prev):
    return ((cur - prev) / prev * 100) if prev else 0


def _crore(v):
    return f"₹{v:,.0f} crore"


@dataclass
class _Carry:
    """State the reference leaks between its chunk sections through function locals."""
    last_text: str | None = None
    basic_eps: float | None = None
    out: list = field(default_factory=list)

    def emit(self, period, suffix, text, chunk_type, primary_value):
        self.out.append({"id": f"icici_{period.lower()}_{suffix}", "text": text, "period": period,
                         "chunk_type": chunk_type, "statement_type": "consolidated",
                         "primary_value": primary_value})


def _profitability(c: _Carry, company, period, results, cur, prev):
    if not (cur and "income" in results and "expenses" in results and "profitAndLoss" in results):
        return
    inc, exp, pnl = results["income"], results["expenses"], results["profitAndLoss"]
    at = lambda table, key, when=cur: table[key].get(when, 0)
    total_income, interest_income, other_income = at(inc, "totalIncome"), at(inc, "interestEarned"), at(inc, "otherIncome")
    total_exp, interest_exp, operating_exp = at(exp, "totalExpenditure"), at(exp, "interestExpended"), at(exp, "operatingExpenses")
    operating_profit, net_profit, provisions = at(pnl, "operatingProfit"), at(pnl, "netProfitForThePeriod"), at(pnl, "provisions")
    prev_income, prev_profit = at(inc, "totalIncome", prev), at(pnl, "netProfitForThePeriod", prev)

    lines = [f"{company} {period} NET PROFIT PROFITABILITY ANALYSIS:\n\n",
             f"NET PROFIT: {_crore(net_profit)}"]
    if prev_profit:
        lines.append(f" ({_growth(net_profit, prev_profit):+.1f}% YoY growth)")
    lines.append(f"\nOperating Profit: {_crore(operating_profit)}")
    lines.append(f"\nNet Margin: {_pct(net_profit, total_income):.1f}% | "
                 f"Operating Margin: {_pct(operating_profit, total_income):.1f}%\n\n")
    lines.append(f"INCOME: Total {_crore(total_income)}")
    if prev_income:
        lines.append(f" ({_growth(total_income, prev_income):+.1f}% YoY)")
    # the reference divides by total_income unguarded here (:138-139)
    lines.append(f"\nInterest Income: {_crore(interest_income)} ({interest_income / total_income * 100:.1f}%)")
    lines.append(f"\nOther Income: {_crore(other_income)} ({other_income / total_income * 100:.1f}%)\n\n")
    lines.append(f"EXPENSES: Total {_crore(total_exp)}")
    lines.append(f"\nInterest: {_crore(interest_exp)} | Operating: {_crore(operating_exp)}")
    lines.append(f"\nProvisions: {_crore(provisions)} | Cost Ratio: {_pct(total_exp, total_income):.1f}%")
    c.last_text = "".join(lines)
    c.emit(period, "profitability_analysis", c.last_text, "profitability_analysis", net_profit)


def _balance_sheet(c: _Carry, company, period, bal, cur):
    if not (bal and "consolidatedBalanceSheet" in bal):
        return
    bs = bal["consolidatedBalanceSheet"]
    if not (bs and "assets" in bs and "capitalAndLiabilities" in bs):
        return
    assets, liab = bs["assets"], bs["capitalAndLiabilities"]
    a = lambda key: assets[key].get(cur, 0)
    l = lambda key: liab[key].get(cur, 0)
    total, advances, investments, cash = a("totalAssets"), a("advances"), a("investments"), a("cashAndBalancesWithRBI")
    deposits, borrowings, capital, reserves = l("deposits"), l("borrowings"), l("capital"), l("reservesAndSurplus")
    text = (f"{company} {period} Balance Sheet Analysis:\n\n"
            f"ASSET COMPOSITION (Total: {_crore(total)}):\n"
            f"• Advances: {_crore(advances)} ({_pct(advances, total):.1f}% of total assets)\n"
            f"• Investments: {_crore(investments)} ({_pct(investments, total):.1f}% of total assets)\n"
            f"• Cash & RBI Balances: {_crore(cash)}\n\n"
            f"FUNDING STRUCTURE:\n"
            f"• Customer Deposits: {_crore(deposits)}\n"
            f"• Borrowings: {_crore(borrowings)}\n"
            f"• Deposit-to-Funding Ratio: {_pct(deposits, deposits + borrowings):.1f}%\n\n"
            f"CAPITAL POSITION:\n"
            f"• Share Capital: {_crore(capital)}\n"
            f"• Reserves & Surplus: {_crore(reserves)}\n"
            f"• Total Equity: {_crore(capital + reserves)}")
    c.last_text = text
    c.emit(period, "balance_sheet_health", text, "balance_sheet_analysis", total)


def _ratios(c: _Carry, company, period, results, cur, prev):
    if "ratios" in results:
        ratios = results["ratios"]
        text = f"{company} {period} Key Financial Ratios & Metrics:\n\n"
        if "earningsPerShare" in ratios:
            eps = ratios["earningsPerShare"]
            basic, diluted, prev_basic = eps["basic"].get(cur, 0), eps["diluted"].get(cur, 0), eps["basic"].get(prev, 0)
            c.basic_eps = basic
            text += f"EARNINGS METRICS:\n• Basic EPS: ₹{basic:.2f} per share"
            if prev_basic:
                text += f" ({_growth(basic, prev_basic):+.1f}% YoY)"
            text += f"\n• Diluted EPS: ₹{diluted:.2f} per share\n\n"
        c.last_text = text
    # reference :231 tests whatever text the function last built
    if c.last_text is not None and len(c.last_text) > 100:
        c.emit(period, "key_ratios", c.last_text, "financial_ratios",
               c.basic_eps if c.basic_eps is not None else 0)


def _segments(c: _Carry, company, period, seg, cur) -> bool:
    """Returns False when the reference would `return chunks` early."""
    payload = None
    if seg:
        if "consolidatedSegmentalResults" in seg:
            payload = seg["consolidatedSegmentalResults"]
        elif "segmentalResults" in seg:
            payload = seg["segmentalResults"]
    if not payload:
        return True
    if "segmentRevenue" in payload and "segmentResults" in payload:
        revenue, profit = payload["segmentRevenue"], payload["segmentResults"]
    elif "segmentRevenue" in payload and "segmentalResults" in payload:
        revenue, profit = payload["segmentRevenue"], payload["segmentalResults"]
    else:
        return False
    rows, total = [], 0
    for key, name in _SEGMENTS:
        if key in revenue and cur in revenue[key]:
            rev = revenue[key][cur]
            res = profit[key].get(cur, 0) if key in profit else 0
            total += rev
            rows.append((name, rev, res, _pct(res, rev)))
    rows.sort(key=lambda r: r[1], reverse=True)   # stable, like list.sort in the reference
    text = f"{company} {period} Retail Banking & Business Segment Performance:\n\n"
    for name, rev, res, margin in rows:
        text += (f"{name.upper()} SEGMENT:\n"
                 f"• Revenue: {_crore(rev)} ({_pct(rev, total):.1f}%)\n"
                 f"• Segment Result: {_crore(res)}\n"
                 f"• Margin: {margin:.1f}%\n\n")
    text += f"TOTAL SEGMENT REVENUE: {_crore(total)}"
    c.last_text = text
    c.emit(period, "segment_performance", text, "segment_analysis", total)
    return True


def create_complete_context_chunks(quarterly_data, period):
    """Same name, arguments and return value as the reference function
    ("chunking_storing (1).py":31): a list of
    {id, text, period, chunk_type, statement_type, primary_value} dicts."""
    fin, seg, bal = classify(quarterly_data)
    c = _Carry()
    if not fin:
        return c.out
    company = fin.get("company", "ICICI Bank Limited")
    cur, prev = period_keys(period)
    if "consolidatedResults" in fin:
        _profitability(c, company, period, fin["consolidatedResults"], cur, prev)
    _balance_sheet(c, company, period, bal, cur)
    if "consolidatedResults" in fin:
        _ratios(c, company, period, fin["consolidatedResults"], cur, prev)
    _segments(c, company, period, seg, cur)
    return c.out


def load_quarter(folder: str):
    docs = []
    for path in glob.glob(os.path.join(folder, "*.json")):   # reference: unsorted glob (:355)
        try:
            with open(path, "r") as f:
                docs.append(json.load(f))
        except Exception:
            continue
    return docs


def build_all_chunks(data_folder: str = "extract_data"):
    """The reference's driver loop (:335-374): four quarters in order."""
    chunks = []
    for quarter in QUARTERS:
        folder = os.path.join(data_folder, f"icici_{quarter}")
        if os.path.exists(folder):
            chunks.extend(create_complete_context_chunks(load_quarter(folder), PERIOD_OF[quarter]))
    return chunks


def insert_columns(chunks, embeddings):
    """The seven parallel columns of the reference's collection.insert (:383-392)."""
    return [[c["id"] for c in chunks], [c["text"] for c in chunks], embeddings,
            [c["period"] for c in chunks], [c["chunk_type"] for c in chunks],
            [c["statement_type"] for c in chunks], [c["primary_value"] for c in chunks]]
